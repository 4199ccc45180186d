#!/usr/bin/env python3
"""Golden vectors for the per-pair evaluation helpers BY RUNNING THE REFERENCE ITSELF.

Build container only (needs /root/reference).  Imports the reference's ``torch_find_matches`` / ``warp_keypoints``
(utils/preprocess_utils.py) and ``compute_pixel_error`` / ``pose_auc`` (utils/common.py) unmodified -- ``cv2`` is absent and
stubbed by tools/_ref_stubs, none of the four functions calls it -- and stores their outputs on seeded inputs from the
portable generator ``gims_amd.synth`` in ``tests/golden/eval_*.npz``.  Nothing from the reference's source is written.  (utils/common.py as a whole needs
torchvision / matplotlib, which are not installed: its two functions are compiled from the file by name.)

    python tools/gen_golden_eval.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "_ref_stubs"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import ast  # noqa: E402

from utils.preprocess_utils import torch_find_matches, warp_keypoints  # noqa: E402  (the reference)
from gims_amd import synth  # noqa: E402


def _reference_functions(path, names):
    """utils/common.py cannot be imported here (torchvision, matplotlib, ... are not installed), so the requested
    top-level functions are compiled straight from the reference's file -- its own code, executed, not copied."""
    tree = ast.parse(open(path).read())
    ns = {"np": np, "torch": torch}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


compute_pixel_error, pose_auc = _reference_functions("/root/reference/utils/common.py", ["compute_pixel_error", "pose_auc"])

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    # ---- GT matching on homography pairs: (n, seed, noise px, dist_thresh, n_iters)
    for n, seed, noise, thr, iters in [(300, 3000, 0.5, 3, 3), (1024, 3001, 0.5, 3, 3), (700, 3002, 1.5, 3, 1), (512, 3003, 2.5, 3, 3),
                                       (2048, 3004, 0.7, 3, 3)]:
        pair, H = synth.make_homography_pair(n, seed, pos_noise=noise)
        k0, k1 = torch.from_numpy(pair["keypoints0"][0]), torch.from_numpy(pair["keypoints1"][0])
        if seed == 3002:
            k1 = k1[:500]                       # unequal counts
        ma0, ma1, mi0, mi1 = torch_find_matches(k0, k1, torch.from_numpy(H), dist_thresh=thr, n_iters=iters)
        warped = warp_keypoints(k0, torch.from_numpy(H))
        np.savez_compressed(os.path.join(OUT, f"eval_gt_n{n}_s{seed}.npz"), meta=np.array([n, seed, thr, iters, len(k1)]),
                            noise=np.float64(noise), ma0=ma0.numpy(), ma1=ma1.numpy(), miss0=mi0.numpy(), miss1=mi1.numpy(),
                            warped=warped.numpy())
        print(f"eval_gt n={n} seed={seed}: {len(ma0)} GT matches, {len(mi0)} / {len(mi1)} unmatched")
    # ---- pixel error and AUC on seeded numbers
    r = np.random.default_rng(7)
    a, b = r.normal(size=(4, 2)).astype(np.float32) * 50, r.normal(size=(4, 2)).astype(np.float32) * 50
    errs = [np.abs(r.normal(size=k)) * s for k, s in ((199, 6.0), (10, 30.0), (50, 1.0), (3, 100.0))]
    np.savez_compressed(os.path.join(OUT, "eval_metrics.npz"), pa=a, pb=b, pixel_error=np.float64(compute_pixel_error(a, b)),
                        **{f"errors{i}": e for i, e in enumerate(errs)},
                        **{f"auc{i}": np.asarray(pose_auc(list(e), [5, 10, 25])) for i, e in enumerate(errs)})
    print("eval_metrics written")


if __name__ == "__main__":
    main()
