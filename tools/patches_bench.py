#!/usr/bin/env python3
"""Measurement of the patch-extraction stage: Gaussian pyramid of one image + one 64x64x3 bicubic affine warp per
keypoint + the 2x2 area halving, on the device (csrc/patches.hip).  The reference does this on one CPU core through OpenCV
(README.md:151,155: 3.2-3.9 s per image at ~15 k keypoints).

    python tools/patches_bench.py [--kpts 8192] [--h 672] [--w 896] [--reps 5]

Prints ONE JSON line: keypoints/s (image and keypoints resident in HBM), the split pyramid / warp times, the algorithmic HBM
bytes of the warp (16 taps x 3 bytes read per patch pixel from the cached level, 12 KiB written per patch) and the CPU
oracle (oracle/patch_oracle.py, NumPy) on a bounded sample."""
import argparse
import json
import os
import sys
import time
from collections import namedtuple

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KP = namedtuple("KP", "pt size angle response octave")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kpts", type=int, default=8192)
    ap.add_argument("--h", type=int, default=672)
    ap.add_argument("--w", type=int, default=896)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    import __graft_entry__
    __graft_entry__.build()
    from gims_amd import hip
    r = np.random.default_rng(0)
    yy, xx = np.mgrid[0:a.h, 0:a.w].astype(np.float32)
    img = np.clip(128 + 50 * np.sin(xx / 9)[..., None] * np.cos(yy / 7)[..., None] + r.normal(0, 12, (a.h, a.w, 3)), 0, 255).astype(np.uint8)
    kp4 = np.stack([r.uniform(0, a.w, a.kpts), r.uniform(0, a.h, a.kpts), r.uniform(2, 10, a.kpts), r.uniform(0, 360, a.kpts)], 1).astype(np.float32)
    octv = ((r.integers(-1, 3, a.kpts) & 0xFF) | (r.integers(0, 4, a.kpts) << 8)).astype(np.int32)
    dimg, dkp, doct = torch.from_numpy(img).cuda(), torch.from_numpy(kp4).cuda(), torch.from_numpy(octv).cuda()

    def run():
        pyr, levels, dev_levels = hip.pyramid_build(dimg)
        return hip.patch_extract(pyr, dev_levels, len(levels), dkp, doct)[0]

    for _ in range(2):
        out = run()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t_pyr = t_warp = 0.0
    for _ in range(a.reps):
        e[0].record()
        pyr, levels, dev_levels = hip.pyramid_build(dimg)
        e[1].record()
        out = hip.patch_extract(pyr, dev_levels, len(levels), dkp, doct)[0]
        e[2].record()
        torch.cuda.synchronize()
        t_pyr += e[0].elapsed_time(e[1]); t_warp += e[1].elapsed_time(e[2])
    t_pyr, t_warp = t_pyr / a.reps, t_warp / a.reps
    res = {"metric": "patch extraction: keypoints/sec (image and keypoints resident in HBM)", "value": a.kpts / ((t_pyr + t_warp) * 1e-3), "unit": "keypoints/s",
           "keypoints": a.kpts, "image": [a.h, a.w, 3], "pyramid_levels": len(levels), "ms_pyramid": t_pyr, "ms_warp": t_warp,
           "warp": {"written_bytes": a.kpts * 32 * 32 * 3 * 4, "gathered_bytes": a.kpts * 64 * 64 * 16 * 3, "note": "gathers hit the cached pyramid level"},
           "patch_mean": float(out.mean())}
    if not a.no_cpu:
        from oracle import patch_oracle as P
        t0 = time.perf_counter()
        pyr_ref = P.build_pyramid(img)
        t1 = time.perf_counter()
        n = 64
        P.compute_patches([(kp4[i, 0], kp4[i, 1], kp4[i, 2], kp4[i, 3], int(octv[i])) for i in range(n)], pyr_ref)
        t2 = time.perf_counter()
        res["cpu_baseline"] = {"value": 1.0 / ((t1 - t0) / a.kpts + (t2 - t1) / n), "unit": "keypoints/s", "cores": 1, "kind": "port",
                               "sample": f"oracle/patch_oracle.py (NumPy): pyramid {t1 - t0:.1f} s once per image, {n} patches in {t2 - t1:.1f} s"}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
