"""gims_eval_pairs (csrc/eval.hip, through the C ABI) against the eval oracle and the reference's golden vectors."""
import numpy as np
import pytest
import torch

from gims_amd import synth
from tests.helpers import golden_names, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hipmod():
    from gims_amd import hip
    hip.load()
    return hip


def _items(hip, specs):
    """specs: list of (kp0, kp1, matches0, mscores0, H, h, w) NumPy arrays -> eval items on the GPU."""
    items = []
    for kp0, kp1, m0, s0, H, h, w in specs:
        n0 = len(kp0)
        items.append(dict(kpts0=torch.from_numpy(kp0).cuda(), kpts1=torch.from_numpy(kp1).cuda(), matches0=torch.from_numpy(m0).cuda(),
                          mscores0=torch.from_numpy(s0).cuda(), h_gt=H, height=h, width=w,
                          gt0=torch.empty(n0, dtype=torch.int32, device="cuda"), inlier=torch.empty(n0, dtype=torch.uint8, device="cuda"),
                          record=torch.zeros(16, device="cuda"), homographies=torch.zeros(18, device="cuda")))
    return items


@pytest.mark.parametrize("name", golden_names("eval_gt_"))
def test_gt_matching_equals_reference_golden(hipmod, name):
    """GT correspondences: exactly the index sets the reference's torch_find_matches returned."""
    g = load_golden(name)
    n, seed, thr, iters, n1 = [int(x) for x in g["meta"]]
    pair, H = synth.make_homography_pair(n, seed, pos_noise=float(g["noise"]))
    kp0, kp1 = pair["keypoints0"][0], np.ascontiguousarray(pair["keypoints1"][0][:n1])
    m0 = np.full(n, -1, dtype=np.int64)
    it = _items(hipmod, [(kp0, kp1, m0, np.zeros(n, np.float32), H, 240, 320)])
    hipmod.eval_pairs(it, dist_thresh=thr, n_iters=iters, ransac_iters=0)
    gt0 = it[0]["gt0"].cpu().numpy()
    ref = np.full(n, -1, dtype=np.int64)
    ref[g["ma0"]] = g["ma1"]
    np.testing.assert_array_equal(gt0, ref)
    rec = it[0]["record"].cpu().numpy()
    assert rec[1] == len(g["ma0"]) and rec[0] == 0 and rec[10] == 0 and rec[9] == 0


def test_records_vs_oracle_batched(hipmod):
    """Precision / recall / 4-point homography / RANSAC / corner errors of a ragged batch against the CPU oracle."""
    from oracle import eval_oracle as E
    specs, refs = [], []
    for n, seed, noise, out_frac in [(400, 3200, 0.4, 0.1), (1024, 3201, 0.6, 0.2), (300, 3202, 1.0, 0.3)]:
        pair, H = synth.make_homography_pair(n, seed, pos_noise=noise, outlier_frac=out_frac)
        kp0, kp1, gt = pair["keypoints0"][0], pair["keypoints1"][0], pair["gt_perm"]
        # a plausible matcher output: most planted correspondences, some wrong, some missing
        r = np.random.default_rng(seed)
        m0 = np.where(gt >= 0, gt, -1).astype(np.int64)
        wrong = r.random(n) < 0.1
        m0[wrong] = r.integers(0, n, size=int(wrong.sum()))
        m0[r.random(n) < 0.1] = -1
        s0 = r.random(n).astype(np.float32)
        w, h = synth.canvas_for(n)
        specs.append((kp0, kp1, m0, s0, H, h, w))
        ma0, ma1, _, _ = E.find_gt_matches(torch.from_numpy(kp0), torch.from_numpy(kp1), torch.from_numpy(H), 3, 3)
        prec, rec, gtv = E.precision_recall(m0, ma0, ma1)
        valid = m0 > -1
        mk0, mk1, mc = kp0[valid], kp1[m0[valid]], s0[valid]
        Hd = E.dlt_top4(mk0, mk1, mc)
        Hr, mask = E.ransac_homography(mk0, mk1, seed=99, iters=500, thresh=3.0)
        refs.append(dict(prec=prec, rec=rec, gt=gtv, err_dlt=E.corner_error(Hd, H, h, w), err_ransac=E.corner_error(Hr, H, h, w),
                         n_in=int(mask.sum()), mask=mask, valid=valid, Hd=Hd, Hr=Hr))
    items = _items(hipmod, specs)
    hipmod.eval_pairs(items, dist_thresh=3, n_iters=3, ransac_thresh=3.0, ransac_iters=500, seed=99)
    for it, ref in zip(items, refs):
        rec = it["record"].cpu().numpy()
        np.testing.assert_array_equal(it["gt0"].cpu().numpy(), ref["gt"])
        assert rec[4] == pytest.approx(ref["prec"], abs=1e-6) and rec[5] == pytest.approx(ref["rec"], abs=1e-6)
        hom = it["homographies"].cpu().numpy().reshape(2, 3, 3)
        np.testing.assert_allclose(hom[0], ref["Hd"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(hom[1], ref["Hr"], rtol=1e-4, atol=1e-5)
        assert rec[7] == pytest.approx(ref["err_dlt"], rel=1e-3, abs=1e-3)
        assert rec[8] == pytest.approx(ref["err_ransac"], rel=1e-3, abs=1e-3)
        inl = it["inlier"].cpu().numpy().astype(bool)
        # same RANSAC specification, float64 on both sides: the inlier sets may differ only for matches sitting on the threshold
        assert (inl[ref["valid"]] != ref["mask"]).sum() <= 2 and abs(rec[6] - ref["n_in"]) <= 2
        assert not inl[~ref["valid"]].any()
        assert rec[8] < 1.0                               # the planted homography is recovered to sub-pixel corner error


def test_end_to_end_matcher_then_eval(hipmod):
    """Matcher output -> evaluation -> AUC, all on the device path; the synthetic pairs carry a planted homography."""
    from gims_amd import GMatcher, evalh
    from tests.helpers import pair_to_data
    m = GMatcher({}).eval()
    m.load_state_dict(synth.make_state_dict(123))
    pairs = [synth.make_homography_pair(512, 3300 + i) for i in range(4)]
    datas = [pair_to_data(p, 15, 2, 7, device="cuda") for p, _ in pairs]
    outs = m.match_pairs(datas)
    ev = evalh.evaluate_pairs(datas, outs, [H for _, H in pairs], ransac_iters=500, seed=5)
    rec = ev["records"].cpu().numpy()
    summ = evalh.summarize(rec)
    assert summ["n_pairs"] == 4
    assert summ["precision"] > 90 and summ["recall"] > 80, summ
    assert summ["auc_ransac"][0] > 80 and rec[:, 8].max() < 1.0, (summ, rec[:, 8])
