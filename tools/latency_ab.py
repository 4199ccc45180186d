#!/usr/bin/env python3
"""Single-pair forward() latency, attention_precision 'auto' against 'bf16', with the per-stage GPU times of one call (diagnostic)."""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.build()
from gims_amd import GMatcher, synth
torch.set_grad_enabled(False)
kpts = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
pair = synth.make_pair(kpts, 1000)
def data():
    d = {k: torch.from_numpy(v).cuda() for k, v in pair.items() if k not in ("gt_perm", "image0", "image1")}
    d["image0"], d["image1"] = pair["image0"], pair["image1"]
    d.update(device=torch.device("cuda"), radius=15, percentile=2, min_size=7)
    return d
for ap in ("auto", "bf16", "auto", "bf16"):
    m = GMatcher({"attention_precision": ap}).eval()
    m.load_state_dict(synth.make_state_dict(123))
    base = data()
    ts = []
    for i in range(14):
        dd = dict(base)
        torch.cuda.synchronize(); t0 = time.perf_counter(); m(dd); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    m.enable_timing(True)
    m(dict(base)); torch.cuda.synchronize()
    st = {k: round(float(np.sum(v)), 3) for k, v in m.stage_times_ms().items()}
    print(ap, "median ms", round(float(np.median(ts[4:])), 3), "stages", st, "sum", round(sum(st.values()), 3), flush=True)
