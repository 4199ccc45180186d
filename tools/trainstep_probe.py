#!/usr/bin/env python3
"""Error distribution of one HIP training step against the trainstep_* fixtures of the reference (run on the GPU box)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gims_amd import GMatcher, synth  # noqa: E402
from tests.helpers import golden_names, grad_sample_index, load_golden, train_data, train_pairs  # noqa: E402

PREC = sys.argv[1] if len(sys.argv) > 1 else "bf16x6"
for name in golden_names("trainstep_"):
    g = load_golden(name)
    m = GMatcher({"train_precision": PREC, "sinkhorn_iterations": int(g["meta"][4]), "pos_loss_weight": float(g["pos_loss_weight"]), "neg_loss_weight": float(g["neg_loss_weight"])})
    m.load_state_dict(synth.make_state_dict(123))
    m = m.cuda().train()
    pairs = train_pairs(name, g)
    for rep in range(2):
        m.zero_grad()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss, pos, neg = m(train_data(pairs, g, device="cuda"), mode="train")
        torch.cuda.synchronize(); t1 = time.perf_counter()
        loss.backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
    grads = {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters()}
    keys = [k[2:] for k in g if k.startswith("g:") or k.startswith("s:")]
    big = max(float(np.abs(g[("g:" if "g:" + k in g else "s:") + k]).max()) for k in keys)
    errs = []
    for k in keys:
        mine = grads[k].astype(np.float64).reshape(-1)
        if "g:" + k in g:
            ref = g["g:" + k].astype(np.float64)
        else:
            ref = g["s:" + k].astype(np.float64)
            mine = mine[grad_sample_index(k, mine.size)]
        errs.append((float(np.abs(mine - ref).max()) / max(float(np.abs(ref).max()), 1e-3 * big), k))
    errs.sort()
    e = np.asarray([x[0] for x in errs])
    print(f"{name}: loss {float(loss):.6f} (ref {float(g['loss']):.6f}); forward {1e3 * (t1 - t0):.1f} ms backward {1e3 * (t2 - t1):.1f} ms; "
          f"grad err median {np.median(e):.2e} p95 {np.quantile(e, 0.95):.2e} worst {errs[-1][0]:.2e} {errs[-1][1]}", flush=True)
    print("     ", [(f"{a:.1e}", b) for a, b in errs[-4:]])
