#!/usr/bin/env python3
"""GPU probe: the CSR mean aggregation of GraphSAGE (gims_sage_mean_split) on 65 536 nodes of mean degree 9, 256 and 128 channels."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gims_amd import hip
hip.load()
rng = np.random.default_rng(0)
n = 65536
deg = rng.integers(3, 16, n); indptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
# neighbours near the node (like a radius graph inside an image of 4096 nodes)
idx = np.concatenate([(i // 4096) * 4096 + rng.integers(0, 4096, d) for i, d in enumerate(deg)]).astype(np.int32)
ip, ix = torch.from_numpy(indptr).cuda(), torch.from_numpy(idx).cuda()
for c in (256, 128):
    h = torch.randn(n, c, device="cuda")
    out = torch.empty(n, 2 * c, dtype=torch.bfloat16, device="cuda")
    for _ in range(3): hip.sage_mean_split(h, ip, ix, out)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): hip.sage_mean_split(h, ip, ix, out)
    b.record(); torch.cuda.synchronize()
    print(f"c={c}: {a.elapsed_time(b) / 20 * 1e3:.1f} us per launch of {n} nodes, mean degree {deg.mean():.1f}")
