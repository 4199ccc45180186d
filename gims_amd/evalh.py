"""Per-pair evaluation after the matcher -- host mirror of the reference's eval loop body (eval_homography.py:186-259).

The reference, for every image pair: matches with ``Matching``; finds ground-truth correspondences from the known
homography (``torch_find_matches``, utils/preprocess_utils.py:98-132); precision / recall (:222-226); a 4-point homography
from the most confident matches and a RANSAC homography (:216-218, OpenCV); the mean corner error of both against the
ground truth (:219-223); finally ``pose_auc`` over all pairs (:237-259, utils/common.py:500-512).

Here the per-pair part is ONE batched call into the kernel library (``gims_eval_pairs``, csrc/eval.hip) on the tensors
``GMatcher.match_pairs`` returns, with no host round trip; the per-pair records are what the ranks all-gather
(``gims_amd.shard``), and the AUC is computed from the gathered records on the host (a few hundred numbers)."""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch

from . import hip

RECORD_FIELDS = hip.EVAL_FIELDS          # columns of the per-pair record (float32): see include/gims_hip.h GIMS_EVAL_*


def evaluate_pairs(datas: Sequence[dict], outs: Sequence[dict], h_gts: Sequence[np.ndarray], dist_thresh: float = 3, n_iters: int = 3,
                   ransac_thresh: float = 3.0, ransac_iters: int = 3000, seed: int = 0) -> dict:
    """datas: the dicts ``match_pairs`` / ``forward`` mutated (kept keypoints, image shapes); outs: the per-pair results;
    h_gts: 3x3 ground-truth homographies mapping image 0 to image 1.  Returns device tensors:
    records [P, 16] float32 (RECORD_FIELDS in the first columns), gt0 / inlier (lists of per-pair tensors),
    homographies [P, 2, 3, 3] float32 (4-point, RANSAC).  Asynchronous on the current stream."""
    P = len(outs)
    dev = outs[0]["matches0"].device
    records = torch.zeros((P, 16), dtype=torch.float32, device=dev)
    homs = torch.zeros((P, 18), dtype=torch.float32, device=dev)
    n0s = [int(o["matches0"].shape[-1]) for o in outs]
    gt_all = torch.empty(sum(n0s), dtype=torch.int32, device=dev)
    in_all = torch.empty(sum(n0s), dtype=torch.uint8, device=dev)
    items, gts, inl, c = [], [], [], 0
    for p, (d, o, H) in enumerate(zip(datas, outs, h_gts)):
        k0 = d["keypoints0"][0].contiguous().float()
        k1 = d["keypoints1"][0].contiguous().float()
        shape0 = d["image0"].shape
        # eval_homography.py:210 takes image0.shape[0] / [1] of the HxWx3 array it loaded; the matcher's dict holds the
        # 1xHxWx3 tensor, hence indices 1 and 2
        height, width = (int(shape0[1]), int(shape0[2])) if len(shape0) == 4 else (int(shape0[0]), int(shape0[1]))
        g, m = gt_all[c:c + n0s[p]], in_all[c:c + n0s[p]]
        c += n0s[p]
        items.append(dict(kpts0=k0, kpts1=k1, matches0=o["matches0"].reshape(-1), mscores0=o["matching_scores0"].reshape(-1).float(),
                          h_gt=H, height=height, width=width, gt0=g, inlier=m, record=records[p], homographies=homs[p]))
        gts.append(g)
        inl.append(m)
    keep = hip.eval_pairs(items, dist_thresh, n_iters, ransac_thresh, ransac_iters, seed)
    return dict(records=records, homographies=homs.view(P, 2, 3, 3), gt0=gts, inlier=inl, _keep=(keep, items))


def pose_auc(errors, thresholds=(5, 10, 25)) -> List[float]:
    """Area under the recall-vs-error curve up to each threshold, as the reference reports it (utils/common.py:500-512)."""
    e = np.sort(np.asarray(list(errors), dtype=np.float64))
    if len(e) == 0:
        return [float("nan") for _ in thresholds]
    rec = np.concatenate([[0.0], (np.arange(len(e)) + 1) / len(e)])
    e = np.concatenate([[0.0], e])
    out = []
    for t in thresholds:
        k = int(np.searchsorted(e, t))
        x = np.concatenate([e[:k], [t]])
        y = np.concatenate([rec[:k], [rec[k - 1]]])
        out.append(float(np.sum((x[1:] - x[:-1]) * (y[1:] + y[:-1]) * 0.5) / t))
    return out


def summarize(records: np.ndarray, min_matches: int = 12, thresholds=(5, 10, 25)) -> dict:
    """records: [P, >= 11] array (host) of per-pair records, e.g. gathered from all ranks.  Mirrors
    eval_homography.py:211-215, 237-259: pairs with fewer than `min_matches` matches (or without a model) are left out;
    AUC in percent for the 4-point and the RANSAC homography, mean precision / recall in percent."""
    r = np.asarray(records, dtype=np.float64)
    col = {k: i for i, k in enumerate(RECORD_FIELDS)}
    ok = (r[:, col["n_valid"]] >= min_matches) & (r[:, col["dlt_ok"]] > 0) & (r[:, col["ransac_ok"]] > 0)
    r = r[ok]
    auc_d = [100.0 * a for a in pose_auc(r[:, col["err_dlt"]], thresholds)]
    auc_r = [100.0 * a for a in pose_auc(r[:, col["err_ransac"]], thresholds)]
    return dict(n_pairs=int(ok.sum()), auc_dlt=auc_d, auc_ransac=auc_r,
                precision=float(100.0 * r[:, col["precision"]].mean()) if len(r) else float("nan"),
                recall=float(100.0 * r[:, col["recall"]].mean()) if len(r) else float("nan"),
                mean_inliers=float(r[:, col["n_inliers"]].mean()) if len(r) else float("nan"))
