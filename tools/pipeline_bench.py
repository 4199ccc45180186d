#!/usr/bin/env python3
"""BASELINE config 5 end to end on one GPU: CAR-HyNet descriptors for both images of a pair (32x32x3 patches around 8192
keypoints each) -> 128-d descriptors duplicated to 256-d (utils/common.py:891) -> the GIMS matcher.  Synthetic inputs:
the keypoints / scores of gims_amd.synth.make_pair and synthetic patches (the second image's patches are the first's,
permuted like its keypoints, plus noise), so the matcher sees correlated descriptors it can actually match.

    python tools/pipeline_bench.py [--kpts 8192] [--pairs 2] [--reps 3]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


PEAK_BF16_TFLOPS = 2500.0
CARHYNET_MFLOP_PER_PATCH = 84.5          # flop counter on the reference's CAR_HyNet, one 32x32x3 patch


def measure_from_images(kpts=8192, pairs=2, reps=3):
    """The same chain starting one stage earlier (SURVEY row f4 inside the timed region): a uint8 image + SIFT-style keypoints (position,
    size, angle, packed octave / layer) per image, resident in HBM -> Gaussian pyramid + one 64x64 affine warp per keypoint + 2x2 halving
    (gims_patch_extract) -> CAR-HyNet -> GMatcher.match_pairs.  Image 1 of a pair is image 0 itself with the keypoints permuted like the
    matcher's synthetic pair (identity homography), so its patches -- and descriptors -- correlate with image 0's.  Detection (OpenCV SIFT in
    the reference) stays outside: keypoints are inputs."""
    from gims_amd import GMatcher, hip, synth
    from gims_amd.carhynet import CARHyNet
    torch.set_grad_enabled(False)
    net = CARHyNet().eval()
    net.load_state_dict(synth.make_carhynet_state_dict(321))
    matcher = GMatcher({}).eval()
    matcher.load_state_dict(synth.make_state_dict(123))
    dev = "cuda"
    work = []
    for p in range(pairs):
        pair = synth.make_pair(kpts, 1000 + p)
        w, h = synth.canvas_for(kpts)
        r = np.random.default_rng(70 + p)
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
        img = np.clip(128 + 50 * np.sin(xx / 9)[..., None] * np.cos(yy / 7)[..., None] + r.normal(0, 12, (h, w, 3)), 0, 255).astype(np.uint8)
        perm = pair["gt_perm"]
        size0, ang0 = r.uniform(2, 10, kpts).astype(np.float32), r.uniform(0, 360, kpts).astype(np.float32)
        oct0 = ((r.integers(-1, 3, kpts) & 0xFF) | (r.integers(0, 4, kpts) << 8)).astype(np.int32)
        kp0 = np.concatenate([pair["keypoints0"][0], size0[:, None], ang0[:, None]], 1).astype(np.float32)
        kp1 = np.empty_like(kp0); oct1 = np.empty_like(oct0)
        kp1[perm] = kp0; oct1[perm] = oct0
        kp1[:, :2] = pair["keypoints1"][0]                      # (image 0's keypoints, permuted and jittered by half a pixel)
        d = {k: torch.from_numpy(v).to(dev) for k, v in pair.items() if k not in ("gt_perm", "image0", "image1", "descriptors0", "descriptors1")}
        d["image0"], d["image1"] = pair["image0"], pair["image1"]
        d.update(device=torch.device(dev), radius=15, percentile=2, min_size=7)
        timg = torch.from_numpy(img).to(dev)
        work.append((d, timg, [torch.from_numpy(k).to(dev) for k in (kp0, kp1)], [torch.from_numpy(o).to(dev) for o in (oct0, oct1)], perm))

    def patches():
        out = []
        for _, timg, kps, octs, _ in work:
            pyr, levels, dev_levels = hip.pyramid_build(timg)     # (both images of the pair are this image: one pyramid)
            out.append([hip.patch_extract(pyr, dev_levels, len(levels), kps[i], octs[i])[0] for i in range(2)])
        return out

    def descriptors(pt):
        return [[net._forward_nhwc(x)[0] for x in pp] for pp in pt]

    def match(descs):
        datas = []
        for (d, _, _, _, _), (d0, d1) in zip(work, descs):
            dd = dict(d)
            dd["descriptors0"], dd["descriptors1"] = torch.cat([d0, d0], 1).t()[None], torch.cat([d1, d1], 1).t()[None]   # common.py:891
            datas.append(dd)
        return matcher.match_pairs(datas)

    def timed(fn, *args):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*args)
        torch.cuda.synchronize()
        return r, time.perf_counter() - t0

    for _ in range(3):
        outs = match(descriptors(patches()))
    t_pat = t_all = 0.0
    for _ in range(reps):
        _, dt = timed(patches); t_pat += dt
        outs, dt = timed(lambda: match(descriptors(patches()))); t_all += dt
    m0 = outs[0]["matches0"][0].cpu().numpy()
    hh, ww = work[0][1].shape[:2]
    return {"metric": f"image-pairs/sec at 2x{kpts} keypoints INCLUDING patch extraction and CAR-HyNet descriptors", "value": pairs * reps / t_all, "unit": "pairs/s",
            "pairs_per_step": pairs, "steps": reps, "ms_per_pair": 1e3 * t_all / (pairs * reps), "ms_per_pair_patches_only": 1e3 * t_pat / (pairs * reps),
            "matches_pair0": int((m0 >= 0).sum()), "data": "synthetic",
            "config": {"workload": f"{pairs} pairs/step: uint8 image {hh}x{ww}x3 + 2x{kpts} keypoints with size / angle / octave -> "
                                   "Gaussian pyramid + 64x64 affine warps + halving (gims_patch_extract) -> CAR-HyNet -> GMatcher.match_pairs; "
                                   "keypoint detection is an input (OpenCV SIFT in the reference)"}}


def measure(kpts=8192, pairs=2, reps=3):
    """The JSON block of one measurement (also embedded in bench.py's line as also["pipeline_2x8192"])."""
    from gims_amd import GMatcher, synth
    from gims_amd.carhynet import CARHyNet
    torch.set_grad_enabled(False)
    net = CARHyNet().eval()
    net.load_state_dict(synth.make_carhynet_state_dict(321))
    matcher = GMatcher({}).eval()
    matcher.load_state_dict(synth.make_state_dict(123))
    dev = "cuda"
    work = []
    for p in range(pairs):
        pair = synth.make_pair(kpts, 1000 + p)
        p0 = synth.make_patches(kpts, 50 + p)
        perm = pair["gt_perm"]                                   # keypoint i of image 0 is keypoint perm[i] of image 1
        p1 = np.empty_like(p0)
        p1[perm] = np.clip(p0 + 0.01 * np.random.default_rng(p).normal(size=p0.shape).astype(np.float32), 0, 1)
        d = {k: torch.from_numpy(v).to(dev) for k, v in pair.items() if k not in ("gt_perm", "image0", "image1", "descriptors0", "descriptors1")}
        d["image0"], d["image1"] = pair["image0"], pair["image1"]
        d.update(device=torch.device(dev), radius=15, percentile=2, min_size=7)
        work.append((d, torch.from_numpy(p0).to(dev), torch.from_numpy(p1).to(dev), perm))

    def descriptors():
        return [[net._forward_nhwc(pt)[0] for pt in (p0, p1)] for _, p0, p1, _ in work]

    def match(descs):
        datas = []
        for (d, _, _, _), (d0, d1) in zip(work, descs):
            dd = dict(d)
            dd["descriptors0"], dd["descriptors1"] = torch.cat([d0, d0], 1).t()[None], torch.cat([d1, d1], 1).t()[None]   # common.py:891
            datas.append(dd)
        return matcher.match_pairs(datas)

    def timed(fn, *args):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*args)
        torch.cuda.synchronize()
        return r, time.perf_counter() - t0

    for _ in range(3):                                           # (the first matcher call measures the attention peakedness at bf16x3)
        outs = match(descriptors())
    t_desc = t_match = t_all = 0.0
    for _ in range(reps):
        descs, dt = timed(descriptors); t_desc += dt
        outs, dt = timed(match, descs); t_match += dt
        outs, dt = timed(lambda: match(descriptors())); t_all += dt
    m0 = outs[0]["matches0"][0].cpu().numpy()
    perm = work[0][3]
    n_patches = 2 * kpts * pairs * reps
    ch_tflops = n_patches * CARHYNET_MFLOP_PER_PATCH * 1e6 / t_desc / 1e12
    return {"metric": f"image-pairs/sec at 2x{kpts} keypoints INCLUDING CAR-HyNet descriptor extraction", "value": pairs * reps / t_all,
            "unit": "pairs/s", "pairs_per_step": pairs, "steps": reps, "ms_per_pair": 1e3 * t_all / (pairs * reps),
            "ms_per_pair_descriptors_only": 1e3 * t_desc / (pairs * reps), "ms_per_pair_matcher_only": 1e3 * t_match / (pairs * reps),
            "matches_pair0": int((m0 >= 0).sum()), "correct_vs_planted_pair0": int(((m0 >= 0) & (m0 == perm[:len(m0)])).sum()),
            "data": "synthetic", "dtype": "split-bf16x3 MFMA convolutions (f32-class) + the matcher's dtypes",
            "config": {"workload": f"{pairs} pairs/step: 32x32x3 patches of 2x{kpts} keypoints -> CAR-HyNet 128-d descriptors (duplicated to 256-d) -> "
                                   "GMatcher.match_pairs (BASELINE config 5)"},
            "carhynet": {"patches_per_s": n_patches / t_desc, "bound": "mfma", "achieved": ch_tflops, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": ch_tflops / PEAK_BF16_TFLOPS, "mfma_passes": 3, "mfma_issue_frac": 3 * ch_tflops / PEAK_BF16_TFLOPS,
                         "algorithmic_mflop_per_patch": CARHYNET_MFLOP_PER_PATCH,
                         "note": "whole descriptor network (7 convolutions + SandGlass + FRN/CoordAtt blocks), wall time of the launches between "
                                 "two device synchronisations"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kpts", type=int, default=8192)
    ap.add_argument("--pairs", type=int, default=2)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    import __graft_entry__
    __graft_entry__.build()
    res = measure(a.kpts, a.pairs, a.reps)
    res["from_images"] = measure_from_images(a.kpts, a.pairs, a.reps)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
