#!/usr/bin/env python3
"""cProfile of the HOST side of GMatcher.match_pairs at 2x1024 x 32 pairs (the configuration whose step is host-bound): where the
Python time of a step goes.  Diagnostic; run on a GPU box."""
import cProfile, pstats, sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.build()
from gims_amd import GMatcher, synth
torch.set_grad_enabled(False)
kpts, pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 32
m = GMatcher({}).eval()
m.load_state_dict(synth.make_state_dict(123))
inputs = []
for pid in range(pairs):
    pair = synth.make_pair(kpts, 1000 + pid)
    d = {k: torch.from_numpy(v).cuda() for k, v in pair.items() if k not in ("gt_perm", "image0", "image1")}
    d["image0"], d["image1"] = pair["image0"], pair["image1"]
    d.update(device=torch.device("cuda"), radius=15, percentile=2, min_size=7)
    inputs.append(d)
def step():
    return m.match_pairs([dict(d) for d in inputs])
for _ in range(4):
    step()
torch.cuda.synchronize()
ts = []
for _ in range(10):
    t0 = time.perf_counter(); step(); ts.append(1e3 * (time.perf_counter() - t0))
torch.cuda.synchronize()
print("host ms per step:", " ".join(f"{x:.2f}" for x in ts))
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
