#!/usr/bin/env python3
"""GPU probe: N forward() calls of ONE pair (B = 1) at a given keypoint count -- run it under
`rocprofv3 --kernel-trace --stats` to see per-kernel durations of the single-pair path.   python3 tools/b1_loop.py 1024 20"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from gims_amd import GMatcher, synth
from helpers import pair_to_data
n, reps = int(sys.argv[1]), int(sys.argv[2])
m = GMatcher({}).eval(); m.load_state_dict(synth.make_state_dict(123))
pair = synth.make_pair(n, 1000)
ts = []
for rep in range(reps):
    data = pair_to_data(pair, 15, 2, 7, device="cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m(data)
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print(f"n={n}: median {sorted(ts)[len(ts)//2]:.2f} ms")
