"""Shared helpers for the parity tests (test infrastructure)."""
import glob
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def golden_names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def pair_to_data(pair, radius, percentile, min_size, device="cpu"):
    d = {k: torch.from_numpy(np.asarray(v)).to(device) for k, v in pair.items()
         if k not in ("gt_perm", "image0", "image1")}
    d["image0"], d["image1"] = pair["image0"], pair["image1"]
    d.update(device=torch.device(device), radius=radius, percentile=percentile, min_size=min_size)
    return d
