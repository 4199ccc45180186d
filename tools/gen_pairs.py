"""Ground-truth rows of data['matches'] for a synthetic pair, the way train.py:113-125 builds them from torch_find_matches:
(b, i0, i1) for correspondences, (b, i0, -1) for image-0 points without partner, (b, -1, i1) for image-1 points without."""
import numpy as np


def matches_of(b, gt_perm, n1):
    i = np.arange(len(gt_perm))
    pos = gt_perm >= 0
    miss1 = np.setdiff1d(np.arange(n1), gt_perm[pos])
    rows = [np.stack([i[pos], gt_perm[pos]], 1), np.stack([i[~pos], -np.ones((~pos).sum(), np.int64)], 1),
            np.stack([-np.ones(len(miss1), np.int64), miss1], 1)]
    m = np.concatenate(rows).astype(np.int64)
    return np.concatenate([np.full((len(m), 1), b, np.int64), m], 1)
