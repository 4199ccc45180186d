#!/usr/bin/env python3
"""Largest move of the Sinkhorn potentials from their start values (u0 = -max(alpha, row max), v0 = 0) over a solve, per workload: what an
adaptive re-derivation rule of the on-chip kernel has to tell apart (dense pairs: no mid-solve derivation needed; sparse pairs: needed)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.build()
from gims_amd import GMatcher, synth
torch.set_grad_enabled(False)
os.environ["GIMS_OT_RESIDENT"] = "0"
def data(pair, r=15, p=2, ms=7):
    d = {k: torch.from_numpy(v).cuda() for k, v in pair.items() if k not in ("gt_perm", "image0", "image1")}
    d["image0"], d["image1"] = pair["image0"], pair["image1"]
    d.update(device=torch.device("cuda"), radius=r, percentile=p, min_size=ms)
    return d
for iters in (100, 20):
    m = GMatcher({"sinkhorn_iterations": iters}).eval()
    m.load_state_dict(synth.make_state_dict(123))
    alpha = float(m.state_dict()["bin_score"])
    cases = [("dense 256", synth.make_pair(256, 1002)), ("dense 1024", synth.make_pair(1024, 1000)), ("dense 4096", synth.make_pair(4096, 1000)),
             ("unbalanced 1500/900", synth.make_pair_unbalanced(1500, 900, 700, 3001)), ("unbalanced 300/520", synth.make_pair_unbalanced(300, 520, 200, 3002)),
             ("sparse 512@800x600", synth.make_pair(512, 2000, canvas=(800, 600))), ("sparse 1024@800x600", synth.make_pair(1024, 2001, canvas=(800, 600)))]
    for name, pair in cases:
        m(data(pair))
        it = m._last["items"][0]
        n, mm = it["n"], it["m"]
        z = it["scores"][:, :mm].float().cpu().numpy()
        uv = it["uv"].cpu().numpy()
        u, v = uv[:n + 1], uv[n + 1:n + mm + 2]
        u0 = -np.maximum(alpha, z.max(axis=1))
        du = u[:n] - u0
        print(f"I={iters:3d} {name:22s} n={n:5d} m={mm:5d}  u-u0 [{du.min():6.1f}, {du.max():6.1f}]  u_bin+alpha {u[n] + alpha:6.1f}  v [{v[:mm].min():6.1f}, {v[:mm].max():6.1f}]  v_bin {v[mm]:6.1f}")
