#!/usr/bin/env python3
"""Generate golden vectors for the GIMS matcher hot path BY RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference).  The reference's
``models/gmatcher.py`` / ``models/agc.py`` are imported unmodified; three third-party
modules that are not installed here (dgl, torch_scatter, cv2) are provided by
``tools/_ref_stubs`` (see the docstrings there -- the DGL pieces are restated from DGL's
documented semantics, so that part is "parity unpinned" against a real DGL wheel).

Outputs: small ``.npz`` fixtures under ``tests/golden/`` holding inputs (or the seeds of the
portable generator ``gims_amd.synth``) and the reference's outputs.  Nothing from the
reference's source is written anywhere.

    python tools/gen_golden.py            # regenerate every fixture
"""
import contextlib
import io
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "_ref_stubs"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from models import gmatcher as RG  # noqa: E402  (the reference)
from models import agc as RA  # noqa: E402  (the reference)
from gims_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_grad_enabled(False)


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def ref_model(sd, config):
    m = RG.GMatcher(dict(config)).eval()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return m


def to_data(pair, radius, percentile, min_size):
    d = {k: torch.from_numpy(v) for k, v in pair.items() if k not in ("gt_perm", "image0", "image1")}
    d["image0"], d["image1"] = pair["image0"], pair["image1"]      # NumPy NHWC, as the callers pass
    d.update(device=torch.device("cpu"), radius=radius, percentile=percentile, min_size=min_size)
    return d


def edges_of(nxg):
    e = np.asarray([(min(u, v), max(u, v)) for u, v in nxg.edges], dtype=np.int64).reshape(-1, 2)
    return e[np.lexsort((e[:, 1], e[:, 0]))]


def agc_stages(pair, side, radius, percentile, min_size):
    """Run the reference AGC stage by stage (agc.py:694-698) and record every intermediate."""
    kp = torch.from_numpy(pair["keypoints" + side])
    de = torch.from_numpy(pair["descriptors" + side])
    sc = torch.from_numpy(pair["scores" + side])
    with quiet():
        g = RA.fast_build_graph_with_cosine_similarity(kp, de, sc, radius, percentile)[0]
        coarse = edges_of(g)
        descs = de[0].permute(1, 0).numpy()
        sim = RA.fast_cosine_similarity_matrix(descs)
        vals = sim[np.triu_indices_from(sim, k=1)]
        thr = RA.fast_percentile_threshold(vals, percentile)
        g = RA.connect_isolated_nodes(g)
        iso = edges_of(g)
        g, kept = RA.remove_small_components(g, min_size)
        kept = np.asarray(sorted(kept), dtype=np.int64)
        g = RA.fast_connect_components(g)
        final = edges_of(g)
    # margin of the closest radius-candidate similarity to the threshold (conditioning of the edge test)
    from scipy.spatial import cKDTree
    cand = np.asarray(sorted(cKDTree(pair["keypoints" + side][0]).query_pairs(r=radius)), dtype=np.int64).reshape(-1, 2)
    margin = float(np.min(np.abs(sim[cand[:, 0], cand[:, 1]] - thr))) if len(cand) else float("inf")
    return {"coarse": coarse, "iso": iso, "kept": kept, "final": final, "thr": np.float32(thr),
            "n_cand": np.int64(len(cand)), "margin": np.float64(margin)}


def run_reference(model, pair, radius, percentile, min_size, capture=False):
    data = to_data(pair, radius, percentile, min_size)
    cap = {}
    hooks = []
    if True:
        hooks.append(model.gnn_encoder.register_forward_hook(lambda m, i, o: cap.setdefault("sage", []).append(o.clone())))
        hooks.append(model.kenc.register_forward_hook(lambda m, i, o: cap.setdefault("kenc", []).append(o.clone())))
        hooks.append(model.gnn.register_forward_hook(lambda m, i, o: cap.__setitem__("gnn", (o[0].clone(), o[1].clone()))))
        for li in (0, 1, 17):
            hooks.append(model.gnn.layers[li].register_forward_hook(
                lambda m, i, o, li=li: cap.setdefault(f"delta{li}", []).append(o.clone())))
        orig = RG.log_optimal_transport

        def spy(scores, alpha, iters):
            cap["scores"] = scores.clone()
            z = orig(scores, alpha, iters)
            cap["ot"] = z.clone()
            return z
        RG.log_optimal_transport = spy
    try:
        with quiet():
            out = model(data)
    finally:
        for h in hooks:
            h.remove()
        RG.log_optimal_transport = orig
    res = {
        "kept0": np.asarray(data["kept_kpts0_indices"][0], dtype=np.int64),
        "kept1": np.asarray(data["kept_kpts1_indices"][0], dtype=np.int64),
        "matches0": out["matches0"][0].numpy().astype(np.int64),
        "matches1": out["matches1"][0].numpy().astype(np.int64),
        "matching_scores0": out["matching_scores0"][0].numpy(),
        "matching_scores1": out["matching_scores1"][0].numpy(),
    }
    for s in ("0", "1"):
        g = data["graph" + s][0]
        src, dst = g.edges()
        res["dgl_src" + s], res["dgl_dst" + s] = src.numpy(), dst.numpy()
    inner = cap["ot"][0][:-1, :-1]
    t0 = inner.topk(2, dim=1).values
    t1 = inner.topk(2, dim=0).values
    # conditioning of the mutual argmax: top-1 / top-2 gap of every row / column of the OT matrix
    res["gap0"] = (t0[:, 0] - t0[:, 1]).numpy()
    res["gap1"] = (t1[0] - t1[1]).numpy()
    if capture:
        res.update(sage0=cap["sage"][0].numpy(), sage1=cap["sage"][1].numpy(),
                   kenc0=cap["kenc"][0][0].numpy(), kenc1=cap["kenc"][1][0].numpy(),
                   gnn0=cap["gnn"][0][0].numpy(), gnn1=cap["gnn"][1][0].numpy(),
                   scores=cap["scores"][0].numpy(), ot=cap["ot"][0].numpy(),
                   mdesc0=out["mdesc0"].numpy(), mdesc1=out["mdesc1"].numpy())
        for li in (0, 1, 17):
            res[f"delta{li}_0"], res[f"delta{li}_1"] = cap[f"delta{li}"][0][0].numpy(), cap[f"delta{li}"][1][0].numpy()
    return res


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def main():
    sd = synth.make_state_dict(123)
    # fingerprint of the portable generator (so the GPU box can verify it reproduces the same bits)
    fp = {k: np.float64(np.asarray(v, dtype=np.float64).sum()) for k, v in sd.items()
          if k in ("final_proj.weight", "gnn.layers.17.mlp.0.weight", "kenc.encoder.0.weight",
                   "gnn_encoder.layers.0.fc_neigh.weight", "gnn.layers.3.mlp.1.running_var")}
    p = synth.make_pair(64, 1000)
    save("synth_fingerprint", **{"sd/" + k: v for k, v in fp.items()},
         kpts0=p["keypoints0"], desc0_sum=np.float64(p["descriptors0"].astype(np.float64).sum()),
         kpts1=p["keypoints1"], gt_perm=p["gt_perm"], scores1=p["scores1"])

    only_ln = "--only-ln" in sys.argv
    model100 = ref_model(sd, {})                                                    # GMatcher defaults
    model20 = ref_model(sd, {"sinkhorn_iterations": 20, "match_threshold": 0.02})   # eval-script setting

    # ---- full pipeline, with intermediates (small) -------------------------------------------
    for n, seed in (() if only_ln else ((64, 1000), (200, 1001))):
        pair = synth.make_pair(n, seed, canvas=synth.canvas_for(256) if n == 200 else None)
        r = run_reference(model100, pair, 15, 2, 7, capture=True)
        arrs = {"in/" + k: v for k, v in pair.items()}
        arrs.update({"out/" + k: v for k, v in r.items()})
        for s in ("0", "1"):
            st = agc_stages(pair, s, 15, 2, 7)
            arrs.update({f"agc{s}/" + k: v for k, v in st.items()})
        arrs["meta"] = np.asarray([n, seed, 15, 2, 7, 100], dtype=np.int64)
        arrs["match_threshold"] = np.float64(0.2)
        save(f"full_n{n}_s{seed}", **arrs)

    # ---- end-to-end outputs only (inputs regenerated from the seed) ---------------------------
    for n, seed, (rad, pct, ms), mdl, iters, thr in () if only_ln else (
            (256, 1002, (15, 2, 7), model100, 100, 0.2),
            (256, 1003, (25, 7, 8), model100, 100, 0.2),          # GMatcher default AGC params
            (512, 1004, (15, 2, 7), model20, 20, 0.02),           # eval-script setting
            (1024, 1000, (15, 2, 7), model100, 100, 0.2),         # BASELINE config 2 shape
            (1024, 1001, (15, 2, 7), model20, 20, 0.02)):
        pair = synth.make_pair(n, seed)
        r = run_reference(mdl, pair, rad, pct, ms)
        arrs = {"out/" + k: v for k, v in r.items()}
        for s in ("0", "1"):
            st = agc_stages(pair, s, rad, pct, ms)
            arrs.update({f"agc{s}/" + k: v for k, v in st.items()})
        arrs["meta"] = np.asarray([n, seed, rad, pct, ms, iters], dtype=np.int64)
        arrs["match_threshold"] = np.float64(thr)
        save(f"e2e_n{n}_s{seed}_r{rad}p{pct}m{ms}_i{iters}", **arrs)

    # ---- use_layernorm=True (gmatcher.py:19-20, 74-85): the LayerNorm variant of every MLP --------
    sd_ln = synth.make_state_dict(123, use_layernorm=True)
    model_ln = ref_model(sd_ln, {"use_layernorm": True})
    for n, seed in ((256, 1005), (1024, 1006)):
        pair = synth.make_pair(n, seed)
        r = run_reference(model_ln, pair, 15, 2, 7)
        arrs = {"out/" + k: v for k, v in r.items()}
        arrs["meta"] = np.asarray([n, seed, 15, 2, 7, 100], dtype=np.int64)
        arrs["match_threshold"] = np.float64(0.2)
        save(f"lne2e_n{n}_s{seed}_r15p2m7_i100", **arrs)
    if "--only-ln" in sys.argv:
        return

    # ---- AGC-only cases that exercise isolated-node fix-up, component removal and linking ------
    for n, seed, canvas, (rad, pct, ms) in (
            (512, 2000, (800, 600), (15, 2, 7)),       # sparse: most points dropped
            (1024, 2001, (800, 600), (15, 2, 7)),
            (1024, 2002, (800, 600), (25, 7, 8)),
            (300, 2003, (200, 150), (15, 50, 5)),      # p=50: many edges cut -> many components
            (2048, 2004, (640, 480), (15, 2, 7))):
        pair = synth.make_pair(n, seed, canvas=canvas)
        arrs = {}
        for s in ("0", "1"):
            st = agc_stages(pair, s, rad, pct, ms)
            arrs.update({f"agc{s}/" + k: v for k, v in st.items()})
        arrs["meta"] = np.asarray([n, seed, rad, pct, ms, canvas[0], canvas[1]], dtype=np.int64)
        save(f"agc_n{n}_s{seed}_r{rad}p{pct}m{ms}", **arrs)

    # ---- closed-form / known-answer vectors computed by the reference functions ----------------
    sc = torch.tensor([[[2.0, 0.0], [0.0, 1.0]]])
    z = RG.log_optimal_transport(sc, torch.tensor(1.0), 100)
    kp = torch.tensor([[[0.0, 0.0], [320.0, 240.0], [160.0, 120.0]]])
    nk = RG.normalize_keypoints(kp, (1, 240, 320, 3))
    rng = np.random.default_rng(7)
    sc2 = torch.from_numpy(rng.normal(size=(1, 37, 53)).astype(np.float32) * 3)
    z2 = RG.log_optimal_transport(sc2, torch.tensor(0.5), 20)
    q = torch.from_numpy(rng.normal(size=(1, 64, 4, 33)).astype(np.float32))
    k = torch.from_numpy(rng.normal(size=(1, 64, 4, 45)).astype(np.float32))
    v = torch.from_numpy(rng.normal(size=(1, 64, 4, 45)).astype(np.float32))
    att, _ = RG.attention(q, k, v)
    save("known_answers", ot2x2=z[0].numpy(), nk_in=kp[0].numpy(), nk_out=nk[0].numpy(),
         ot_in=sc2[0].numpy(), ot_alpha=np.float32(0.5), ot_out=z2[0].numpy(),
         att_q=q[0].numpy(), att_k=k[0].numpy(), att_v=v[0].numpy(), att_out=att[0].numpy(),
         pct_len=np.asarray([523776, 8386560, 33550336, 10, 1], dtype=np.int64),
         pct_k=np.asarray([int(L * 2 / 100) for L in (523776, 8386560, 33550336)] + [min(int(10 * 100 / 100), 9), 0],
                          dtype=np.int64))


if __name__ == "__main__":
    main()
