#!/usr/bin/env python3
"""GPU probe: per-phase cycles of the 2-D on-chip Sinkhorn kernel (GIMS_OT_PROF=1) for a few geometries."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.ot_probe import make, run
from gims_amd import hip
hip.load()
os.environ["GIMS_OT_RESIDENT"] = "2"
cases = [(int(a), int(b)) for a, b in (s.split("x") for s in sys.argv[1:])] or [(4096, 2), (1022, 32), (2000, 6)]
for n, np_ in cases:
    items = make(n, np_)
    for wt in ("0", "1"):
        os.environ["GIMS_OT_PROF"] = "0"
        t, _, _ = run(items, 100, True)
        print(f"n={n} x{np_}: {t:.3f} ms per call", flush=True)
        os.environ["GIMS_OT_PROF"] = "1"
        run(items, 100, True, reps=1)
        break
