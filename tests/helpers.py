"""Shared helpers for the parity tests (test infrastructure)."""
import glob
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def make_rare_pair(n, seed, n_hot, gain):
    """The input of the raree2e_* fixtures (tools/gen_golden_rare.py): synth.make_pair(n, seed) in which `n_hot` keypoints of image 0 and their
    partners in image 1 carry descriptors scaled by `gain` -- a few sharply peaked attention rows inside otherwise diffuse layers."""
    from gims_amd import synth
    pair = synth.make_pair(n, seed)
    rng = np.random.default_rng(seed + 7)
    hot0 = np.sort(rng.choice(n, n_hot, replace=False))
    hot1 = pair["gt_perm"][hot0]
    pair["descriptors0"] = pair["descriptors0"].copy()
    pair["descriptors1"] = pair["descriptors1"].copy()
    pair["descriptors0"][0][:, hot0] *= np.float32(gain)          # (1, D, N) channel-major, like the reference's callers pass them
    pair["descriptors1"][0][:, hot1] *= np.float32(gain)
    return pair, hot0, hot1


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


def compare_with_golden(out, data, g, thr, score_tol=1e-4):
    """One pair's outputs (result dict + the mutated data dict) against an e2e_* fixture of the reference: kept ids, EVERY match index of both
    images, int64 / f32 dtypes, scores within 1e-4 (BASELINE.json's bars)."""
    ids = lambda k: k.cpu().numpy() if torch.is_tensor(k) else np.asarray(k)       # noqa: E731  (match_pairs leaves them on the device)
    np.testing.assert_array_equal(ids(data["kept_kpts0_indices"][0]), g["out/kept0"])
    np.testing.assert_array_equal(ids(data["kept_kpts1_indices"][0]), g["out/kept1"])
    m0, m1 = out["matches0"][0].cpu().numpy(), out["matches1"][0].cpu().numpy()
    s0, s1 = out["matching_scores0"][0].cpu().numpy(), out["matching_scores1"][0].cpu().numpy()
    assert out["matches0"].dtype == torch.int64 and out["matching_scores0"].dtype == torch.float32
    r0, r1, rs0, rs1 = g["out/matches0"], g["out/matches1"], g["out/matching_scores0"], g["out/matching_scores1"]
    safe0 = (g["out/gap0"] > 1e-3) & (np.abs(rs0 - thr) > 1e-3)
    # a row is also unsafe when its partner column's argmax is ill-conditioned (mutual check)
    col_unsafe = g["out/gap1"] <= 1e-3
    partner = np.where(r0 >= 0, r0, 0)
    safe0 &= ~col_unsafe[partner] | (r0 < 0)
    assert safe0.mean() > 0.97, f"fixture too ill-conditioned: {safe0.mean():.3f}"
    bad = np.nonzero((m0 != r0) & safe0)[0]
    assert len(bad) == 0, f"{len(bad)} well-conditioned match indices differ, e.g. rows {bad[:5]}: {m0[bad[:5]]} vs {r0[bad[:5]]}"
    same = m0 == r0
    err = np.abs(s0 - rs0)[same & (r0 >= 0)].max()
    assert err < score_tol, f"matching_scores0 max err {err:.3e}"
    # ... and in practice EVERY row agrees, ill-conditioned ones included: asserted, so that a regression on those rows is
    # seen (a failure here with zero well-conditioned mismatches means a reference decision flipped on a sub-1e-3 margin)
    mismatched_unsafe = int((m0 != r0).sum())
    assert mismatched_unsafe == 0, f"{mismatched_unsafe} ill-conditioned rows differ from the reference: {np.nonzero(m0 != r0)[0][:8]}"
    np.testing.assert_array_equal(m1, r1)
    if score_tol == 1e-4:
        assert np.abs(s1 - rs1).max() < 1e-4                     # every row, matched or not
    else:                                                        # a fixture with its own (documented) bar: matched rows
        assert np.abs(s1 - rs1)[(m1 == r1) & (r1 >= 0)].max() < score_tol
    return dict(n=len(m0), mismatched_unsafe=mismatched_unsafe, score_err=float(err))


def safe_rows(ot, thr, ref_matches0, ref_scores0, eps=1e-3):
    """Rows of the reference's (n+1, m+1) log-OT matrix whose match decision is well conditioned: top-1/top-2 gap of the
    row above `eps`, score further than `eps` from the threshold, and a well-conditioned argmax in the column the row points at (the
    partner's column for matched rows, the best candidate's column for unmatched ones).  Index parity is asserted on exactly these rows; everywhere else a last-ulp difference in the
    potentials may legitimately flip the reference's own decision."""
    inner = np.asarray(ot)[:-1, :-1]
    part = np.partition(inner, -2, axis=1)
    gap0 = part[:, -1] - part[:, -2]
    partc = np.partition(inner, -2, axis=0)
    gap1 = partc[-1] - partc[-2]
    safe = (gap0 > eps) & (np.abs(np.asarray(ref_scores0) - thr) > eps)
    r0 = np.asarray(ref_matches0)
    partner = np.where(r0 >= 0, r0, 0)
    safe &= (gap1[partner] > eps) | (r0 < 0)
    # an UNMATCHED row is a decision too: "the column of my best candidate prefers another row" flips with a last-ulp difference when that
    # column's own top two are tied (tests/test_fuzz_vs_oracle_gpu.py case 9: two rows with the bit-identical value -1.3604851 in one column;
    # the reference's argmax takes the first, any other evaluation order may take the second)
    safe &= (gap1[inner.argmax(axis=1)] > eps) | (r0 >= 0)
    return safe


def train_pairs(name, g):
    """The synthetic pairs of a trainloss_* fixture (regenerated from seeds; the fixture stores only outputs)."""
    from gims_amd import synth
    return {"trainloss_n256_s1002_i100": lambda: [synth.make_pair(256, 1002)],
            "trainloss_n1024_s1000_i100": lambda: [synth.make_pair(1024, 1000)],
            "trainloss_n1024sparse_s2001_i20": lambda: [synth.make_pair(1024, 2001, canvas=(800, 600))],
            "trainloss_b2_n64_s1000_i100": lambda: [synth.make_pair(64, 1000), synth.make_pair(64, 1000, desc_noise=0.2)],
            "trainstep_n256_s1002_i100": lambda: [synth.make_pair(256, 1002)],
            "trainstep_n512_s1003_i20": lambda: [synth.make_pair(512, 1003)],
            "trainstep_b2_n64_s1000_i100": lambda: [synth.make_pair(64, 1000), synth.make_pair(64, 1000, desc_noise=0.2)],
            "trainstep_n1024sparse_s2001_i20": lambda: [synth.make_pair(1024, 2001, canvas=(800, 600))],
            "trainstep_ln_n256_s1002_i100": lambda: [synth.make_pair(256, 1002)],
            "trainstep_n2048_s1004_i100": lambda: [synth.make_pair(2048, 1004)],
            "trainstep_n4096_s1005_i20": lambda: [synth.make_pair(4096, 1005)]}[name]()


def train_data(pairs, g, device="cpu"):
    """Batch dict for mode='train' exactly as tools/gen_golden_train.py built it for the reference."""
    datas = [pair_to_data(p, int(g["meta"][1]), int(g["meta"][2]), int(g["meta"][3]), device=device) for p in pairs]
    data = {k: (torch.cat([d[k] for d in datas]) if torch.is_tensor(datas[0][k]) else datas[0][k]) for k in datas[0]}
    data["image0"] = np.concatenate([p["image0"] for p in pairs])
    data["image1"] = np.concatenate([p["image1"] for p in pairs])
    data["matches"] = torch.from_numpy(g["matches"]).to(device)
    return data


def check_score_gradients(g, dscores, dbin, n_pairs, rtol):
    """d loss / d scores and d loss / d bin_score against what the reference's autograd produced (trainloss_* fixtures):
    dense matrices where the fixture holds them, a fixed sample of cells + row / column sums otherwise.  rtol is relative
    to the largest gradient entry of the pair."""
    for b in range(n_pairs):
        d = np.asarray(dscores[b], dtype=np.float64)
        scale = float(g[f"dscores_absmax_{b}"])
        assert scale > 0
        if f"dscores_{b}" in g:
            assert d.shape == g[f"dscores_{b}"].shape
            err = np.abs(d - g[f"dscores_{b}"]).max()
        else:
            err = np.abs(d.reshape(-1)[g[f"dscores_sample_idx_{b}"]] - g[f"dscores_sample_{b}"]).max()
        assert err <= rtol * scale, (b, err, scale)
        np.testing.assert_allclose(d.sum(1), g[f"dscores_rowsum_{b}"], atol=8 * rtol * scale, rtol=0)
        np.testing.assert_allclose(d.sum(0), g[f"dscores_colsum_{b}"], atol=8 * rtol * scale, rtol=0)
    assert abs(float(dbin) - float(g["dbin_score"])) <= rtol * max(abs(float(g["dbin_score"])), 1e-3), (float(dbin), float(g["dbin_score"]))


def grad_sample_index(name: str, numel: int) -> np.ndarray:
    """The stored entries of a large gradient in the trainstep_* fixtures (same rule as tools/gen_golden_grads.py)."""
    import zlib
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    return np.sort(rng.choice(numel, 256, replace=False)).astype(np.int64)


def check_step_gradients(g, grads, rtol, rtol_p95=None, atol_frac=1e-3):
    """``grads``: parameter name -> gradient (NumPy) of one training step; compared with a trainstep_* fixture of the
    reference's own autograd.  Every stored gradient must be present.  Per tensor the error is the largest entry difference
    relative to the largest reference entry of THAT tensor (floor: atol_frac of the largest gradient entry anywhere -- the
    gradients of the key bias, of a bias in front of a BatchNorm and of bin_score are mathematically zero or rounding noise).
    Bars: every tensor within ``rtol``; 95 % of the tensors within ``rtol_p95`` (two f32 evaluations of the same step differ
    by a few 1e-4 on most tensors and by up to a few 1e-3 where one ReLU of a near-zero pre-activation falls the other way:
    measured between the reference and the oracle, tests/test_train_oracle_cpu.py).  Returns (worst, where, p95)."""
    keys = [k[2:] for k in g if k.startswith("g:") or k.startswith("s:")]
    assert keys and all(k in grads for k in keys), [k for k in keys if k not in grads][:5]
    big = max(float(np.abs(g[("g:" if "g:" + k in g else "s:") + k]).max()) for k in keys)
    errs = []
    for k in keys:
        mine = np.asarray(grads[k], dtype=np.float64).reshape(-1)
        assert np.isfinite(mine).all(), k
        if "g:" + k in g:
            ref = g["g:" + k].astype(np.float64)
        else:
            ref = g["s:" + k].astype(np.float64)
            tot, nrm, amax = g["n:" + k]
            den = max(nrm, atol_frac * big * np.sqrt(mine.size))
            assert abs(np.sqrt((mine ** 2).sum()) - nrm) <= rtol * den, (k, np.sqrt((mine ** 2).sum()), nrm)
            mine = mine[grad_sample_index(k, mine.size)]
        den = max(float(np.abs(ref).max()), atol_frac * big)
        errs.append((float(np.abs(mine - ref).max()) / den, k))
    errs.sort()
    worst, p95 = errs[-1], errs[int(0.95 * (len(errs) - 1))][0]
    assert worst[0] <= rtol, errs[-5:]
    if rtol_p95 is not None:
        assert p95 <= rtol_p95, (p95, errs[-20:])
    return worst[0], worst[1], p95
