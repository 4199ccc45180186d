"""Seeded random configurations of the whole path -- unbalanced keypoint counts, every combination of radius / percentile /
min_size the reference's scripts use, both Sinkhorn settings -- through the HIP path and through the CPU oracle (oracle/gims_oracle.py,
the restatement pinned to the reference's goldens by tests/test_oracle_golden.py): kept keypoint sets and graphs' node counts equal,
match indices equal on every well-conditioned row, scores within 1e-4 (BASELINE.json north_star).  The goldens cover the sizes the
reference was run at; this file covers the shapes nobody picked by hand."""
import numpy as np
import pytest
import torch

from gims_amd import GMatcher, synth
from oracle import gims_oracle as O
from tests.helpers import pair_to_data, safe_rows

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

RADII, PERCENTILES, MIN_SIZES = (8, 15, 25), (2, 5, 7), (3, 7, 8)          # agc.py's defaults and the values of eval_homography.py / README


def _case(seed):
    rng = np.random.default_rng(9000 + seed)
    lo, hi = (40, 700) if seed < 20 else (700, 2200)                        # seeds 20+: the sizes where the big-tile kernels take over
    n0, n1 = int(rng.integers(lo, hi)), int(rng.integers(lo, hi))
    common = int(rng.integers(min(n0, n1) // 3, min(n0, n1) + 1))
    side = float(np.sqrt(max(n0, n1)) * rng.uniform(9.0, 16.0))             # density: from "almost everything kept" to "a third removed"
    canvas = (int(side * 1.25) + 8, int(side * 0.8) + 8)
    r, p, ms = (int(rng.choice(x)) for x in (RADII, PERCENTILES, MIN_SIZES))
    iters, thr = ((100, 0.2), (20, 0.02))[seed % 2]
    return dict(n0=n0, n1=n1, common=common, canvas=canvas, r=r, p=p, ms=ms, iters=iters, thr=thr, seed=1300 + seed)


@pytest.fixture(scope="module")
def matchers(synth_sd):
    out = {}
    for iters, thr in ((100, 0.2), (20, 0.02)):
        m = GMatcher({"sinkhorn_iterations": iters, "match_threshold": thr}).eval()
        m.load_state_dict(synth_sd)
        m(pair_to_data(synth.make_pair(256, 1002), 15, 2, 7, device="cuda"))          # 'auto' attention: the calibrating batch
        out[iters] = m
    return out


def _against_oracle(out, d_gpu, c, pair, synth_sd, kept_of=lambda d, s: d["kept_kpts%d_indices" % s][0]):
    d_cpu = pair_to_data(pair, c["r"], c["p"], c["ms"], device="cpu")
    st = {}
    try:
        ref = O.gmatcher_forward(synth_sd, d_cpu, {"sinkhorn_iterations": c["iters"], "match_threshold": c["thr"]}, stages=st)
    except ValueError:
        return None                                                                    # everything removed: covered by its own test
    for s in (0, 1):
        k = kept_of(d_gpu, s)
        k = k.cpu().tolist() if torch.is_tensor(k) else list(k)
        assert k == list(d_cpu["kept_kpts%d_indices" % s][0]), (c, s)
    r0, rs0 = ref["matches0"][0].numpy(), ref["matching_scores0"][0].numpy()
    m0, s0 = out["matches0"][0].cpu().numpy(), out["matching_scores0"][0].cpu().numpy()
    m1, s1 = out["matches1"][0].cpu().numpy(), out["matching_scores1"][0].cpu().numpy()
    assert m0.shape == r0.shape and m1.shape == ref["matches1"][0].numpy().shape
    safe = safe_rows(st["ot"][0].numpy(), c["thr"], r0, rs0)
    np.testing.assert_array_equal(m0[safe], r0[safe])
    same = m0 == r0
    assert np.abs(s0 - rs0)[same].max(initial=0.0) < 1e-4, c
    same1 = m1 == ref["matches1"][0].numpy()
    assert np.abs(s1 - ref["matching_scores1"][0].numpy())[same1].max(initial=0.0) < 1e-4, c
    return float(safe.mean())


@pytest.mark.parametrize("seed", list(range(16)) + [40, 41, 42, 43])
def test_random_configuration_vs_oracle(matchers, synth_sd, seed):
    c = _case(seed)
    pair = synth.make_pair_unbalanced(c["n0"], c["n1"], c["common"], c["seed"], canvas=c["canvas"])
    d_gpu = pair_to_data(pair, c["r"], c["p"], c["ms"], device="cuda")
    try:
        out = matchers[c["iters"]](d_gpu)
    except ValueError:
        with pytest.raises(ValueError):                                                # ... then the oracle must refuse the same way
            O.gmatcher_forward(synth_sd, pair_to_data(pair, c["r"], c["p"], c["ms"], device="cpu"), {})
        return
    frac = _against_oracle(out, d_gpu, c, pair, synth_sd)
    assert frac is not None and frac > 0.7, (c, frac)


def test_random_ragged_batch_vs_oracle(matchers, synth_sd):
    """The same kind of pairs as ONE ragged match_pairs batch (one shared graph setting, as match_pairs requires): every pair against the oracle."""
    cases, pairs, datas = [], [], []
    for seed in (*range(16, 20), *range(28, 34)):
        c = dict(_case(seed), r=15, p=2, ms=7, iters=100, thr=0.2)
        pair = synth.make_pair_unbalanced(c["n0"], c["n1"], c["common"], c["seed"], canvas=c["canvas"])
        try:
            O.gmatcher_forward(synth_sd, pair_to_data(pair, 15, 2, 7, device="cpu"), {})
        except ValueError:
            continue                                                                   # a pair the reference itself refuses cannot sit in a batch
        cases.append(c)
        pairs.append(pair)
        datas.append(pair_to_data(pair, 15, 2, 7, device="cuda"))
    assert len(cases) >= 5
    outs = matchers[100].match_pairs(datas)
    torch.cuda.synchronize()
    fr = [_against_oracle(o, d, c, p, synth_sd) for o, d, c, p in zip(outs, datas, cases, pairs)]
    assert min(fr) > 0.7, fr
