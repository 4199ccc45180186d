"""Pin the CPU oracle (oracle/gims_oracle.py) against golden vectors produced by the reference itself
(tools/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from gims_amd import synth
from oracle import gims_oracle as O
from tests.helpers import check_score_gradients, golden_names, load_golden, pair_to_data, train_data, train_pairs

torch.set_grad_enabled(False)


def test_portable_generator_fingerprint(synth_sd):
    g = load_golden("synth_fingerprint")
    for k in [k for k in g if k.startswith("sd/")]:
        assert np.float64(np.asarray(synth_sd[k[3:]], dtype=np.float64).sum()) == g[k], k
    p = synth.make_pair(64, 1000)
    np.testing.assert_array_equal(p["keypoints0"], g["kpts0"])
    np.testing.assert_array_equal(p["keypoints1"], g["kpts1"])
    np.testing.assert_array_equal(p["gt_perm"], g["gt_perm"])
    np.testing.assert_array_equal(p["scores1"], g["scores1"])
    assert np.float64(p["descriptors0"].astype(np.float64).sum()) == g["desc0_sum"]


def test_known_answers():
    g = load_golden("known_answers")
    z = O.log_optimal_transport(torch.tensor([[[2.0, 0.0], [0.0, 1.0]]]), torch.tensor(1.0), 100)[0]
    np.testing.assert_allclose(z.numpy(), g["ot2x2"], atol=1e-6)
    p = z.exp().numpy()
    np.testing.assert_allclose(p[:2].sum(1), [1, 1], atol=1e-4)      # rows sum to 1,1,(2)
    np.testing.assert_allclose(p[:, :2].sum(0), [1, 1], atol=1e-4)
    nk = O.normalize_keypoints(torch.from_numpy(g["nk_in"])[None], (1, 240, 320, 3))[0]
    np.testing.assert_allclose(nk.numpy(), g["nk_out"], atol=1e-7)
    np.testing.assert_allclose(g["nk_out"], [[-.0067, -.7143], [1.4219, .3571], [.7076, -.1786]], atol=1e-4)
    z2 = O.log_optimal_transport(torch.from_numpy(g["ot_in"])[None], torch.tensor(float(g["ot_alpha"])), 20)[0]
    np.testing.assert_allclose(z2.numpy(), g["ot_out"], atol=1e-6)
    att = O.attention(*[torch.from_numpy(g[k])[None] for k in ("att_q", "att_k", "att_v")])[0]
    np.testing.assert_allclose(att.numpy(), g["att_out"], atol=1e-6)
    for L, k in zip(g["pct_len"][:3], g["pct_k"][:3]):
        assert O.percentile_index(int(L), 2) == int(k)
    assert O.percentile_index(10, 100) == 9 and O.percentile_index(1, 2) == 0


def _check_agc(g, pair, rad, pct, ms):
    for s in ("0", "1"):
        kp = pair["keypoints" + s][0]
        de = pair["descriptors" + s][0].T          # transposed view of (D,N), as agc.py:431
        r = O.agc_build(kp, de, rad, pct, ms)
        assert np.float32(r["threshold"]) == np.float32(g[f"agc{s}/thr"])
        np.testing.assert_array_equal(r["coarse_edges"], g[f"agc{s}/coarse"])
        np.testing.assert_array_equal(r["iso_edges"], g[f"agc{s}/iso"])
        np.testing.assert_array_equal(r["kept"], g[f"agc{s}/kept"])
        np.testing.assert_array_equal(r["final_edges_orig"], g[f"agc{s}/final"])
        assert r["n_candidates"] == int(g[f"agc{s}/n_cand"])


@pytest.mark.parametrize("name", golden_names("agc_"))
def test_agc_only(name):
    g = load_golden(name)
    n, seed, rad, pct, ms, w, h = [int(x) for x in g["meta"]]
    _check_agc(g, synth.make_pair(n, seed, canvas=(w, h)), rad, pct, ms)


@pytest.mark.parametrize("name", golden_names("e2e_"))
def test_e2e(name, synth_sd):
    g = load_golden(name)
    n, seed, rad, pct, ms, iters = [int(x) for x in g["meta"]]
    pair = synth.make_pair(n, seed)
    _check_agc(g, pair, rad, pct, ms)
    data = pair_to_data(pair, rad, pct, ms)
    out = O.gmatcher_forward(synth_sd, data, {"sinkhorn_iterations": iters, "match_threshold": float(g["match_threshold"])})
    np.testing.assert_array_equal(np.asarray(data["kept_kpts0_indices"][0]), g["out/kept0"])
    np.testing.assert_array_equal(np.asarray(data["kept_kpts1_indices"][0]), g["out/kept1"])
    assert out["matches0"].dtype == torch.int64
    np.testing.assert_array_equal(out["matches0"][0].numpy(), g["out/matches0"])
    np.testing.assert_array_equal(out["matches1"][0].numpy(), g["out/matches1"])
    np.testing.assert_allclose(out["matching_scores0"][0].numpy(), g["out/matching_scores0"], atol=5e-5)
    np.testing.assert_allclose(out["matching_scores1"][0].numpy(), g["out/matching_scores1"], atol=5e-5)
    for s in ("0", "1"):   # DGL graph: same directed edge multiset
        gg = data["graph" + s][0]
        deg = np.diff(gg["indptr"])
        dst = np.repeat(np.arange(len(deg)), deg)
        a = np.stack([gg["indices"].astype(np.int64), dst], 1)
        b = np.stack([g["out/dgl_src" + s], g["out/dgl_dst" + s]], 1)
        a = a[np.lexsort((a[:, 1], a[:, 0]))]
        b = b[np.lexsort((b[:, 1], b[:, 0]))]
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("name", [n for n in golden_names("ube2e_") if "n15382" not in n])
def test_e2e_unbalanced(name, synth_sd):
    """The oracle against the reference on UNBALANCED pairs (n0 != n1: image 1 = partners of a subset of image 0's keypoints + fresh outliers,
    synth.make_pair_unbalanced): graph stages of both images, kept ids, matches, scores.  (The 15 382 / 14 870 fixture of the reference's README
    configuration takes the reference four minutes on this CPU and is checked on the GPU only.)"""
    g = load_golden(name)
    n0, n1, nc, seed, rad, pct, ms, iters = [int(x) for x in g["meta"]]
    pair = synth.make_pair_unbalanced(n0, n1, nc, seed)
    assert pair["keypoints0"].shape[1] == n0 != pair["keypoints1"].shape[1] == n1 and int((pair["gt_perm"] >= 0).sum()) == nc
    _check_agc(g, pair, rad, pct, ms)
    data = pair_to_data(pair, rad, pct, ms)
    out = O.gmatcher_forward(synth_sd, data, {"sinkhorn_iterations": iters, "match_threshold": float(g["match_threshold"])})
    np.testing.assert_array_equal(np.asarray(data["kept_kpts0_indices"][0]), g["out/kept0"])
    np.testing.assert_array_equal(np.asarray(data["kept_kpts1_indices"][0]), g["out/kept1"])
    np.testing.assert_array_equal(out["matches0"][0].numpy(), g["out/matches0"])
    np.testing.assert_array_equal(out["matches1"][0].numpy(), g["out/matches1"])
    np.testing.assert_allclose(out["matching_scores0"][0].numpy(), g["out/matching_scores0"], atol=5e-5)
    np.testing.assert_allclose(out["matching_scores1"][0].numpy(), g["out/matching_scores1"], atol=5e-5)
    assert (g["out/matches0"] < 0).any() and (g["out/matches1"] < 0).any()      # rows AND columns end in the dustbin


@pytest.mark.parametrize("name", golden_names("full_"))
def test_full_with_intermediates(name, synth_sd):
    g = load_golden(name)
    n, seed, rad, pct, ms, iters = [int(x) for x in g["meta"]]
    pair = {k[3:]: g[k] for k in g if k.startswith("in/")}
    ref = synth.make_pair(n, seed, canvas=synth.canvas_for(256) if n == 200 else None)
    for k in pair:      # the committed inputs are what the portable generator produces
        np.testing.assert_array_equal(pair[k], ref[k])
    _check_agc(g, pair, rad, pct, ms)
    data = pair_to_data(pair, rad, pct, ms)
    st = {}
    out = O.gmatcher_forward(synth_sd, data, {"sinkhorn_iterations": iters}, stages=st)
    tol = dict(atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(st["sage0"][0].permute(1, 0).numpy(), g["out/sage0"], **tol)
    np.testing.assert_allclose(st["sage1"][0].permute(1, 0).numpy(), g["out/sage1"], **tol)
    np.testing.assert_allclose(st["kenc0"][0].numpy(), g["out/kenc0"], **tol)
    np.testing.assert_allclose(st["kenc1"][0].numpy(), g["out/kenc1"], **tol)
    np.testing.assert_allclose(st["gnn0"][0].numpy(), g["out/gnn0"], atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(st["gnn1"][0].numpy(), g["out/gnn1"], atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(st["scores"][0].numpy(), g["out/scores"], atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(st["ot"][0].numpy(), g["out/ot"], atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(out["mdesc0"].numpy(), g["out/mdesc0"], atol=1e-4, rtol=1e-4)
    np.testing.assert_array_equal(out["matches0"][0].numpy(), g["out/matches0"])
    np.testing.assert_array_equal(out["matches1"][0].numpy(), g["out/matches1"])
    np.testing.assert_allclose(out["matching_scores0"][0].numpy(), g["out/matching_scores0"], atol=5e-5)


@pytest.mark.parametrize("name", golden_names("lne2e_"))
def test_e2e_layernorm(name):
    """use_layernorm=True (gmatcher.py:19-20, 74-85): the oracle's LayerNorm variant against the reference's outputs."""
    g = load_golden(name)
    n, seed, rad, pct, ms, iters = [int(x) for x in g["meta"]]
    pair = synth.make_pair(n, seed)
    data = pair_to_data(pair, rad, pct, ms)
    sd = synth.make_state_dict(123, use_layernorm=True)
    out = O.gmatcher_forward(sd, data, {"sinkhorn_iterations": iters, "match_threshold": float(g["match_threshold"]), "use_layernorm": True})
    np.testing.assert_array_equal(np.asarray(data["kept_kpts0_indices"][0]), g["out/kept0"])
    np.testing.assert_array_equal(out["matches0"][0].numpy(), g["out/matches0"])
    np.testing.assert_array_equal(out["matches1"][0].numpy(), g["out/matches1"])
    np.testing.assert_allclose(out["matching_scores0"][0].numpy(), g["out/matching_scores0"], atol=5e-5)
    np.testing.assert_allclose(out["matching_scores1"][0].numpy(), g["out/matching_scores1"], atol=5e-5)


@pytest.mark.parametrize("name", golden_names("trainloss_"))
def test_train_loss_forward_vs_reference(synth_sd, name):
    """Forward value of forward_train's loss (gmatcher.py:309-386, module in eval mode): oracle == reference."""
    g = load_golden(name)
    pairs = train_pairs(name, g)
    data = train_data(pairs, g)
    cfg = {"sinkhorn_iterations": int(g["meta"][4]), "pos_loss_weight": float(g["pos_loss_weight"]), "neg_loss_weight": float(g["neg_loss_weight"])}
    loss, pos, neg = O.gmatcher_forward(synth_sd, data, cfg, mode="train")
    for b in range(len(pairs)):
        np.testing.assert_array_equal(np.asarray(data["kept_kpts0_indices"][b]), g[f"kept0_{b}"])
    np.testing.assert_allclose([float(loss), float(pos), float(neg)], [g["loss"], g["pos"], g["neg"]], atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("name", [n for n in golden_names("trainloss_") if "n1024_s1000" not in n])
def test_train_loss_score_gradients_vs_reference(synth_sd, name):
    """d loss / d scores and d loss / d bin_score by autograd through the oracle's unrolled Sinkhorn == the reference's autograd."""
    g = load_golden(name)
    pairs = train_pairs(name, g)
    data = train_data(pairs, g)
    cfg = {"sinkhorn_iterations": int(g["meta"][4]), "pos_loss_weight": float(g["pos_loss_weight"]), "neg_loss_weight": float(g["neg_loss_weight"])}
    st = {"grad_scores": True}
    with torch.enable_grad():
        loss, _, _ = O.gmatcher_forward(synth_sd, data, cfg, stages=st, mode="train")
        loss.backward()
    check_score_gradients(g, [st["scores_leaf"].grad[b].numpy() for b in range(len(pairs))], st["alpha_leaf"].grad, len(pairs), rtol=2e-3)
