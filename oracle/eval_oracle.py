"""CPU ORACLE for the per-pair evaluation that follows the matcher -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates, in NumPy + CPU PyTorch, what the reference's eval loop computes for one image pair after ``matching(data)``
returned:

    /root/reference/eval_homography.py:186-226   GT matching, DLT / RANSAC homographies, corner error, precision, recall
    /root/reference/utils/preprocess_utils.py:74-132   torch_cdist, torch_setdiff1d, warp_keypoints, torch_find_matches
    /root/reference/utils/common.py:477-481, 500-512   compute_pixel_error, pose_auc

Only ``tests/`` may import it.  Pinning:
  * ``tools/gen_golden_eval.py`` imports the reference's ``torch_find_matches`` / ``warp_keypoints`` /
    ``compute_pixel_error`` / ``pose_auc`` in the build container and commits their outputs on seeded inputs to
    ``tests/golden/eval_*.npz``; ``tests/test_eval_oracle_golden.py`` checks this file against them (index sets exact).
  * PARITY UNPINNED for the two OpenCV calls of that loop, whose implementation is not in /root/reference and not
    installed here (opencv-python, requirements): ``cv2.getPerspectiveTransform`` (restated as the exact solution of the
    4-point system, which is what it documents) and ``cv2.findHomography(..., cv2.RANSAC)`` (OpenCV's sampler and
    refinement are not reproducible from outside; ``ransac_homography`` below is this build's own, fully specified
    RANSAC: same model, same default 3 px reprojection threshold and 3000 hypotheses = the maxIters the reference passes, eval_homography.py:222; deterministic sampler).
"""
from __future__ import annotations

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


# ------------------------------------------------------------------------------------------------ GT matching
def warp_keypoints(kpts: torch.Tensor, H: torch.Tensor) -> torch.Tensor:
    """preprocess_utils.py:86-96 -- homogeneous warp in float32, same op order as the reference."""
    source = torch.cat([kpts, torch.ones(len(kpts), 1)], dim=-1)
    dest = (H @ source.T).T
    dest = dest / dest[:, 2:3]
    return dest[:, :2]


def find_gt_matches(kp0: torch.Tensor, kp1: torch.Tensor, H: torch.Tensor, dist_thresh: float = 3, n_iters: int = 1):
    """preprocess_utils.py:98-132 (torch_find_matches): iterated mutual nearest neighbours between warp(kp0) and kp1 among
    the still unmatched points, accepted when closer than dist_thresh.  Returns (ma0, ma1, miss0, miss1) index arrays."""
    ma0, ma1 = np.empty(0, np.int64), np.empty(0, np.int64)
    miss0, miss1 = np.arange(len(kp0)), np.arange(len(kp1))
    proj = warp_keypoints(kp0.float(), H.float())
    for _ in range(n_iters):
        a, b = proj[miss0], kp1.float()[miss1]
        if len(a) == 0 or len(b) == 0:
            break
        dist = torch.sqrt(((a[:, None, :] - b[None, :, :]) ** 2).sum(-1))          # torch_cdist, :74-78
        min1, min2 = torch.argmin(dist, 1), torch.argmin(dist, 0)
        j = torch.where(min1[min2] == torch.arange(len(min2)))[0]
        i = min2[j]
        keep = dist[i, j] < dist_thresh
        i, j = i[keep].numpy(), j[keep].numpy()
        m0, m1 = miss0[i], miss1[j]
        miss0, miss1 = np.setdiff1d(miss0, m0), np.setdiff1d(miss1, m1)            # torch_setdiff1d, :80-84 (sorted)
        ma0, ma1 = np.concatenate([ma0, m0]), np.concatenate([ma1, m1])
    return ma0, ma1, miss0, miss1


def precision_recall(matches0: np.ndarray, ma0: np.ndarray, ma1: np.ndarray):
    """eval_homography.py:207-209, 222-226."""
    gt = np.full(len(matches0), -1, dtype=np.int64)
    gt[ma0] = ma1
    valid = matches0 > -1
    match_flag = matches0[ma0] == ma1
    precision = match_flag.sum() / valid.sum()
    fn_flag = np.logical_and(matches0 != gt, matches0 == -1)
    recall = match_flag.sum() / (match_flag.sum() + fn_flag.sum())
    return float(precision), float(recall), gt


# ------------------------------------------------------------------------------------------------ homographies
def perspective_transform(pts: np.ndarray, H: np.ndarray) -> np.ndarray:
    """cv2.perspectiveTransform on an (N, 2) array (documented semantics), float64."""
    p = np.concatenate([pts.astype(np.float64), np.ones((len(pts), 1))], axis=1) @ np.asarray(H, np.float64).T
    return p[:, :2] / p[:, 2:3]


def homography_from_4(src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    """cv2.getPerspectiveTransform (documented semantics): the H with H[2,2] = 1 that maps 4 points exactly."""
    A, b = np.zeros((8, 8)), np.zeros(8)
    for k in range(4):
        x, y = float(src[k, 0]), float(src[k, 1])
        u, v = float(dst[k, 0]), float(dst[k, 1])
        A[2 * k] = [x, y, 1, 0, 0, 0, -u * x, -u * y]
        A[2 * k + 1] = [0, 0, 0, x, y, 1, -v * x, -v * y]
        b[2 * k], b[2 * k + 1] = u, v
    h = np.linalg.solve(A, b)
    return np.append(h, 1.0).reshape(3, 3)


def dlt_top4(mkpts0: np.ndarray, mkpts1: np.ndarray, mconf: np.ndarray) -> np.ndarray:
    """eval_homography.py:216-217: homography through the four most confident matches.  Ties in confidence are broken by
    the lower match index here (NumPy's default argsort is not stable, so the reference leaves them unspecified)."""
    order = np.lexsort((np.arange(len(mconf)), -mconf.astype(np.float64)))[:4]
    return homography_from_4(mkpts0[order], mkpts1[order])


def _splitmix(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return x ^ (x >> np.uint64(31))


def ransac_sample(seed: int, hyp: int, k: int) -> np.ndarray:
    """The sampler shared with the HIP kernel: four DISTINCT indices in [0, k) for hypothesis `hyp` -- successive
    splitmix64 outputs of the state (seed, hyp), each reduced modulo k, duplicates skipped."""
    out, state = [], np.array([(seed ^ (hyp * 0xD1342543DE82EF95)) & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64)
    while len(out) < 4:
        state = _splitmix(state)
        idx = int(state[0] % np.uint64(k))
        if idx not in out:
            out.append(idx)
    return np.asarray(out)


def reproj_error2(H: np.ndarray, p0: np.ndarray, p1: np.ndarray) -> np.ndarray:
    """Squared forward reprojection error ||H p0 - p1||^2 in float64 (points given in float32)."""
    q = perspective_transform(p0, H)
    return ((q - p1.astype(np.float64)) ** 2).sum(-1)


def lsq_homography(p0: np.ndarray, p1: np.ndarray) -> np.ndarray:
    """Least-squares H (H[2,2] = 1) over point pairs: normal equations of the 2K x 8 DLT system, float64."""
    x, y = p0[:, 0].astype(np.float64), p0[:, 1].astype(np.float64)
    u, v = p1[:, 0].astype(np.float64), p1[:, 1].astype(np.float64)
    z, o = np.zeros_like(x), np.ones_like(x)
    A = np.concatenate([np.stack([x, y, o, z, z, z, -u * x, -u * y], 1), np.stack([z, z, z, x, y, o, -v * x, -v * y], 1)])
    b = np.concatenate([u, v])
    h = np.linalg.solve(A.T @ A, A.T @ b)
    return np.append(h, 1.0).reshape(3, 3)


def ransac_homography(p0: np.ndarray, p1: np.ndarray, seed: int, iters: int = 3000, thresh: float = 3.0):
    """This build's RANSAC (see the header): `iters` 4-point hypotheses from `ransac_sample`, score = number of points
    with forward reprojection error <= thresh, best = most inliers (first such hypothesis), then ONE least-squares refit
    on the inliers of the best hypothesis and a final inlier mask under the refit model.  Returns (H, mask) or
    (None, zeros) when fewer than 4 points / no valid hypothesis."""
    k = len(p0)
    if k < 4:
        return None, np.zeros(k, bool)
    best_n, best_H = -1, None
    t2 = float(thresh) ** 2
    for hyp in range(iters):
        s = ransac_sample(seed, hyp, k)
        try:
            H = homography_from_4(p0[s], p1[s])
        except np.linalg.LinAlgError:
            continue
        if not np.isfinite(H).all():
            continue
        with np.errstate(all="ignore"):
            n_in = int((reproj_error2(H, p0, p1) <= t2).sum())
        if n_in > best_n:
            best_n, best_H = n_in, H
    if best_H is None:
        return None, np.zeros(k, bool)
    mask = reproj_error2(best_H, p0, p1) <= t2
    if mask.sum() >= 4:
        try:
            H2 = lsq_homography(p0[mask], p1[mask])
            if np.isfinite(H2).all():
                best_H = H2
                mask = reproj_error2(best_H, p0, p1) <= t2
        except np.linalg.LinAlgError:
            pass
    return best_H, mask


# ------------------------------------------------------------------------------------------------ errors and AUC
def compute_pixel_error(pred_points: np.ndarray, gt_points: np.ndarray) -> float:
    """common.py:477-481."""
    diff = gt_points - pred_points
    return float(np.sqrt((diff ** 2).sum(-1)).mean())


def corner_error(H_est: np.ndarray, H_gt: np.ndarray, height: int, width: int) -> float:
    """eval_homography.py:210, 219-223: mean distance of the four image corners under the two homographies."""
    c = np.array([[0, 0], [0, height], [width, height], [width, 0]], dtype=np.float32)
    return compute_pixel_error(perspective_transform(c, H_est).astype(np.float32), perspective_transform(c, H_gt).astype(np.float32))


def pose_auc(errors, thresholds):
    """common.py:500-512."""
    sort_idx = np.argsort(errors)
    errors = np.array(list(errors))[sort_idx]
    recall = (np.arange(len(errors)) + 1) / len(errors)
    errors = np.r_[0., errors]
    recall = np.r_[0., recall]
    aucs = []
    for t in thresholds:
        last_index = np.searchsorted(errors, t)
        r = np.r_[recall[:last_index], recall[last_index - 1]]
        e = np.r_[errors[:last_index], t]
        aucs.append(np.trapz(r, x=e) / t)
    return aucs
