#!/usr/bin/env python3
"""GPU probe: single-pair forward() latency and batched throughput with the GNN layers replayed as a HIP graph
(non-default stream) (GIMS_OPS_GRAPH=1) vs as a plain recorded sequence (default)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from gims_amd import GMatcher, synth
from helpers import pair_to_data
m = GMatcher({}).eval(); m.load_state_dict(synth.make_state_dict(123))
side = torch.cuda.Stream()
for n, npairs in ((1024, 1), (4096, 1), (1024, 32)):
    pairs = [synth.make_pair(n, 1000 + i) for i in range(npairs)]
    with torch.cuda.stream(side):
        ts = []
        for rep in range(8):
            datas = [pair_to_data(p, 15, 2, 7, device="cuda") for p in pairs]
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = m(datas[0]) if npairs == 1 else m.match_pairs(datas)
            torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    print(f"n={n} x{npairs} GIMS_OPS_GRAPH={os.environ.get('GIMS_OPS_GRAPH', '0')}: ms per call " + " ".join(f"{t:7.2f}" for t in ts))
