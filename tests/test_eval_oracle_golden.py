"""The eval oracle (oracle/eval_oracle.py) against golden vectors produced by the reference's own functions
(tools/gen_golden_eval.py): GT matching index sets exact, warped keypoints bit-exact, pixel error / AUC to 1e-12.
Plus the specification-level properties of the pieces that cannot be pinned (OpenCV is not in /root/reference)."""
import numpy as np
import pytest
import torch

from gims_amd import synth
from tests.helpers import golden_names, load_golden
from oracle import eval_oracle as E


@pytest.mark.parametrize("name", golden_names("eval_gt_"))
def test_gt_matching_equals_reference(name):
    g = load_golden(name)
    n, seed, thr, iters, n1 = [int(x) for x in g["meta"]]
    pair, H = synth.make_homography_pair(n, seed, pos_noise=float(g["noise"]))
    k0, k1 = torch.from_numpy(pair["keypoints0"][0]), torch.from_numpy(pair["keypoints1"][0][:n1])
    np.testing.assert_array_equal(E.warp_keypoints(k0, torch.from_numpy(H)).numpy(), g["warped"])
    ma0, ma1, mi0, mi1 = E.find_gt_matches(k0, k1, torch.from_numpy(H), dist_thresh=thr, n_iters=iters)
    np.testing.assert_array_equal(ma0, g["ma0"])
    np.testing.assert_array_equal(ma1, g["ma1"])
    np.testing.assert_array_equal(mi0, g["miss0"])
    np.testing.assert_array_equal(mi1, g["miss1"])
    # the planted correspondences are what it finds (noise 0.5 px << 3 px): sanity of the fixture itself
    gt = pair["gt_perm"]
    if float(g["noise"]) <= 0.7 and n1 == n:
        planted = gt[ma0] >= 0                   # outlier keypoints may still pair up with a chance neighbour within 3 px
        assert (gt[ma0] == ma1)[planted].mean() > 0.9 and planted.mean() > 0.9


def test_pixel_error_and_auc_equal_reference():
    g = load_golden("eval_metrics")
    assert abs(E.compute_pixel_error(g["pa"], g["pb"]) - float(g["pixel_error"])) < 1e-12
    for i in range(4):
        np.testing.assert_allclose(E.pose_auc(list(g[f"errors{i}"]), [5, 10, 25]), g[f"auc{i}"], rtol=0, atol=1e-12)


def test_precision_recall_formula():
    # 6 keypoints; GT: 0->2, 1->0, 3->3; predictions: 0->2 (right), 1->1 (wrong), 3 unmatched (miss), 4->5 (no GT)
    matches0 = np.array([2, 1, -1, -1, 5, -1])
    p, r, gt = E.precision_recall(matches0, np.array([0, 1, 3]), np.array([2, 0, 3]))
    np.testing.assert_array_equal(gt, [2, 0, -1, 3, -1, -1])
    assert p == pytest.approx(1 / 3) and r == pytest.approx(1 / 2)      # 1 correct of 3 predicted; 1 correct + 1 missed


def test_homography_from_4_and_ransac_recover_planted():
    pair, H = synth.make_homography_pair(600, 3100, pos_noise=0.3, outlier_frac=0.3)
    gt = pair["gt_perm"]
    ok = np.nonzero(gt >= 0)[0]
    p0, p1 = pair["keypoints0"][0][ok], pair["keypoints1"][0][gt[ok]]
    # corrupt a third of the correspondences
    bad = np.arange(0, len(p0), 3)
    p1 = p1.copy()
    p1[bad] = p1[bad[::-1]]
    H4 = E.homography_from_4(p0[[1, 2, 4, 5]], p1[[1, 2, 4, 5]])
    np.testing.assert_allclose(E.perspective_transform(p0[[1, 2, 4, 5]], H4), p1[[1, 2, 4, 5]], atol=1e-6)
    Hr, mask = E.ransac_homography(p0, p1, seed=11, iters=300, thresh=3.0)
    good = np.setdiff1d(np.arange(len(p0)), bad)
    assert mask[good].mean() > 0.98 and mask[bad].mean() < 0.1
    w, h = synth.canvas_for(600)
    assert E.corner_error(Hr, H, h, w) < 0.5
    # sampler: distinct indices, reproducible
    s = E.ransac_sample(11, 5, 7)
    assert len(set(s.tolist())) == 4 and (s == E.ransac_sample(11, 5, 7)).all() and s.max() < 7
