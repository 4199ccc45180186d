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
