#!/usr/bin/env python3
"""How far do the Sinkhorn potentials move from their start values (u0 = -max(alpha, row max), v0 = 0)?  Dense pair vs the sparse pair of
test_sparse_graph_few_kept_vs_oracle, on-chip solve with the re-derivation period at 50 and at 100 against the streamed solve.  Diagnostic."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.build()
from gims_amd import GMatcher, synth, hip
torch.set_grad_enabled(False)
def data(pair):
    d = {k: torch.from_numpy(v).cuda() for k, v in pair.items() if k not in ("gt_perm", "image0", "image1")}
    d["image0"], d["image1"] = pair["image0"], pair["image1"]
    d.update(device=torch.device("cuda"), radius=15, percentile=2, min_size=7)
    return d
m = GMatcher({}).eval()
m.load_state_dict(synth.make_state_dict(123))
alpha = float(m.state_dict()["bin_score"])
for name, pair in (("dense 1024", synth.make_pair(1024, 1000)), ("sparse 512 on 800x600", synth.make_pair(512, 2000, canvas=(800, 600)))):
    res = {}
    for mode, refresh in (("0", "50"), ("2", "50"), ("2", "100")):
        os.environ["GIMS_OT_RESIDENT"], os.environ["GIMS_OT_REFRESH"] = mode, refresh
        m(data(pair)); m(data(pair))
        it = m._last["items"][0]
        n, mm = it["n"], it["m"]
        z = it["scores"][:, :mm].float().cpu().numpy()
        uv = it["uv"].cpu().numpy()
        res[(mode, refresh)] = (uv[:n + 1].copy(), uv[n + 1:n + mm + 2].copy(), it["mscores0"].cpu().numpy().copy())
    u, v, s = res[("0", "50")]
    u0 = -np.maximum(alpha, z.max(axis=1))
    print(f"== {name}: n={n} m={mm} alpha={alpha:.3f}  scores in [{z.min():.1f}, {z.max():.1f}]")
    print(f"   u - u0: min {np.min(u[:n] - u0):.1f} max {np.max(u[:n] - u0):.1f};  u_bin - (-alpha) = {u[n] + alpha:.1f};  v: min {v[:mm].min():.1f} max {v[:mm].max():.1f}; v_bin {v[mm]:.1f}")
    for key in (("2", "50"), ("2", "100")):
        uu, vv, ss = res[key]
        print(f"   on-chip refresh {key[1]}: max |u - streamed| {np.abs(uu - u).max():.2e}, |v - streamed| {np.abs(vv - v).max():.2e}, |score - streamed| {np.abs(ss - s).max():.2e}")

# ---- where does the on-chip solve WITHOUT a mid-solve derivation leave the streamed solve?  iteration sweep on the sparse pair's score matrix
os.environ["GIMS_OT_RESIDENT"], os.environ["GIMS_OT_REFRESH"] = "0", "50"
m(data(synth.make_pair(512, 2000, canvas=(800, 600))))
it0 = m._last["items"][0]
n, mm = it0["n"], it0["m"]
zs = it0["scores"].clone()
def solve(mode, refresh, iters):
    os.environ["GIMS_OT_RESIDENT"], os.environ["GIMS_OT_REFRESH"] = mode, str(refresh)
    item = dict(scores=zs, n=n, m=mm, matches0=torch.empty(n, dtype=torch.int64, device="cuda"), matches1=torch.empty(mm, dtype=torch.int64, device="cuda"),
                mscores0=torch.empty(n, device="cuda"), mscores1=torch.empty(mm, device="cuda"), uv=torch.empty(n + mm + 3, device="cuda"))
    probs = hip.make_ot_problems([item])
    work = torch.empty(hip.sinkhorn_workspace_bytes(probs), dtype=torch.uint8, device="cuda")
    hip.sinkhorn_match(probs, alpha, iters, 0.2, work)
    return item["uv"].cpu().numpy()
print("iters: max |u - streamed|, |v - streamed| of the on-chip solve with refresh -1 (final only) / 50 / 0 (adaptive)")
for iters in (5, 10, 20, 30, 40, 50, 60, 80, 100):
    ref = solve("0", 50, iters)
    row = []
    for refresh in (-1, 50, 0):
        uv = solve("2", refresh, iters)
        row.append(f"{np.abs(uv[:n + 1] - ref[:n + 1]).max():.1e}/{np.abs(uv[n + 1:n + mm + 2] - ref[n + 1:n + mm + 2]).max():.1e}")
    print(f"  {iters:3d}: " + "   ".join(row) + f"   |u| range [{ref[:n+1].min():.1f}, {ref[:n+1].max():.1f}]  v [{ref[n+1:n+mm+2].min():.1f}, {ref[n+1:n+mm+2].max():.1f}]")
