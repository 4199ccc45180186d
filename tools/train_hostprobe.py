#!/usr/bin/env python3
"""GPU probe: is the training step bound by the host or by the GPU?  Per phase: time until the Python call returns (host enqueue) and time
until the device has drained; plus a cProfile of the host side of a few steps."""
import cProfile, os, pstats, sys, time, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gims_amd import GMatcher, synth
from gims_amd.optim import Adam as FusedAdam
from tools.train_bench import batch

cfg = {"sinkhorn_iterations": 100, "pos_loss_weight": 0.45, "neg_loss_weight": 1.0, "train_precision": "bf16x6"}
m = GMatcher(cfg); m.load_state_dict(synth.make_state_dict(123)); m = m.cuda().train()
opt = FusedAdam(m.parameters(), lr=1e-4)
rec = []
prof = cProfile.Profile()
with torch.enable_grad():
    for i in range(14):
        d = batch(2048, 1000 + i % 4, "cuda")
        torch.cuda.synchronize()
        if i == 8:
            prof.enable()
        t0 = time.perf_counter(); loss, pos, neg = m(d, mode="train"); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        loss.backward(); t3 = time.perf_counter()
        torch.cuda.synchronize(); t4 = time.perf_counter()
        opt.step(); opt.zero_grad(); t5 = time.perf_counter()
        torch.cuda.synchronize(); t6 = time.perf_counter()
        if i >= 4 and i < 8:
            rec.append((t1 - t0, t2 - t0, t3 - t2, t4 - t2, t5 - t4, t6 - t4))
prof.disable()
r = np.median(np.asarray(rec), axis=0) * 1e3
print(f"forward: host {r[0]:.2f} ms, drained {r[1]:.2f} ms | backward: host {r[2]:.2f} ms, drained {r[3]:.2f} ms | optimizer: host {r[4]:.2f}, drained {r[5]:.2f} ms")
s = io.StringIO()
pstats.Stats(prof, stream=s).sort_stats(os.environ.get("PROBE_SORT", "tottime")).print_stats(45)
print("\n".join(l[:150] for l in s.getvalue().splitlines()[:75]))
