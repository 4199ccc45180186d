#!/usr/bin/env python3
"""GPU probe: N adaptive-graph builds of 16 images of n keypoints (the graph-build stage of the 4096x8 step on its own) -- run it under
rocprofv3 (--kernel-trace --stats, or --pmc ...) to look at the agc_* kernels.   python3 tools/agc_loop.py [n=4096] [images=16] [reps=5]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import torch
from gims_amd import hip, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
hip.load()
items = []
for b in range(B // 2):
    pair = synth.make_pair(n, 1000 + b)
    for side in ("0", "1"):
        kp = torch.from_numpy(np.ascontiguousarray(pair["keypoints" + side][0])).cuda()
        de = torch.from_numpy(np.ascontiguousarray(pair["descriptors" + side][0].T)).cuda()
        items.append(dict(kpts=kp, desc=de, kept=torch.empty(n, dtype=torch.int32, device="cuda"),
                          indptr=torch.empty(n + 1, dtype=torch.int32, device="cuda"),
                          indices=torch.empty(n * 64, dtype=torch.int32, device="cuda"),
                          info=torch.empty(8, dtype=torch.int32, device="cuda")))
arr = hip.make_agc_images(items)
work = torch.empty(hip.agc_workspace_bytes(arr), dtype=torch.uint8, device="cuda")
ts = []
for r in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hip.agc_build(arr, 15, 2, 7, work)
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print("agc_build ms:", " ".join(f"{t:.3f}" for t in ts), " info[0]:", [int(it["info"][0]) for it in items[:4]], " flags:", [int(it["info"][7]) for it in items[:4]])
