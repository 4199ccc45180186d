#!/usr/bin/env python3
"""GPU probe: latency of ONE pair through the reference-shaped GMatcher.forward (what eval_homography.py does per pair)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from gims_amd import GMatcher, synth
from helpers import pair_to_data
m = GMatcher({}).eval(); m.load_state_dict(synth.make_state_dict(123))
for n in (512, 1024, 2048, 4096, 8192):
    pair = synth.make_pair(n, 1000)
    ts = []
    for rep in range(6):
        data = pair_to_data(pair, 15, 2, 7, device="cuda")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = m(data)
        torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    print(f"n={n}: forward() latency ms (6 calls): " + " ".join(f"{t:7.2f}" for t in ts) + f"   matched {int((out['matches0'] >= 0).sum())}")

# stage anatomy of a single pair (GPU ms per stage, host ms per stage)
for n in (1024, 4096):
    pair = synth.make_pair(n, 1000)
    m.enable_timing(True)
    t_wall = []
    for rep in range(3):
        data = pair_to_data(pair, 15, 2, 7, device="cuda")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m(data)
        torch.cuda.synchronize(); t_wall.append(1e3 * (time.perf_counter() - t0))
    st, sh = m.stage_times_ms(), m.stage_host_ms()
    print(f"n={n} wall {t_wall[-1]:.2f} ms; stage gpu ms (last call): " + "  ".join(f"{k} {sum(v[-(len(v)//3):]):.2f}" for k, v in st.items())
          + f"   SUM {sum(sum(v[-(len(v)//3):]) for v in st.values()):.2f}")
    print("   stage host ms (last call): " + "  ".join(f"{k} {sum(v[-(len(v)//3):]):.2f}" for k, v in sh.items()))
    m.enable_timing(False)
