#!/usr/bin/env python3
"""GPU probe: the three Sinkhorn paths side by side (streamed kernels, 1-D on-chip kernel, 2-D on-chip kernel): time per call and
agreement of potentials / matches with the streamed path.   python tools/ot_probe2.py [4096x8 1022x32 ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.ot_probe import make, run
from gims_amd import hip
hip.load()
cases = [(int(a), int(b)) for a, b in (s.split("x") for s in sys.argv[1:])] or [(1022, 32), (4096, 8), (4096, 2), (4096, 1), (1024, 8), (300, 4), (2000, 6), (1500, 3), (3000, 2)]
for n, np_ in cases:
    items = make(n, np_)
    os.environ["GIMS_OT_RES2"] = "1"
    t0, uv0, m0 = run(items, 100, False)
    os.environ["GIMS_OT_RES2"] = "0"
    t1, uv1, m1 = run(items, 100, 2)
    os.environ["GIMS_OT_RES2"] = "1"
    os.environ["GIMS_OT_R2_INIT"] = "0"
    t3, uv3, m3 = run(items, 100, 2)
    os.environ["GIMS_OT_R2_INIT"] = "1"
    t2, uv2, m2 = run(items, 100, 2)
    for name, (uv, m) in (("1-D", (uv1, m1)), ("2-D", (uv2, m2)), ("2-D, separate init", (uv3, m3))):
        du = max(float((a[:-1] - b[:-1]).abs().max()) for a, b in zip(uv0, uv))
        st = max(float(b[-1]) for b in uv)
        same = all(torch.equal(a, b) for a, b in zip(m0, m))
        print(f"n={n} x{np_} {name}: max|du,dv| {du:.2e} status {st} matches equal {same}", flush=True)
    print(f"n={n} x{np_}: streamed {t0:8.3f} ms   on-chip 1-D {t1:8.3f} ms   on-chip 2-D {t2:8.3f} ms (separate init sweep: {t3:8.3f})", flush=True)
