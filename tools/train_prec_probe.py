#!/usr/bin/env python3
"""GPU probe: gradient error of the training step against the reference goldens, and the step time at 2 x 2048, under the experimental
per-group precisions of the reverse pass (GIMS_TRAIN_PREC_ATTN_BWD / _WGRAD / _AGRAD = x3 | x6)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gims_amd import GMatcher, synth
from tests.helpers import check_step_gradients, golden_names, load_golden, train_data, train_pairs

torch.set_grad_enabled(True)
names = golden_names("trainstep_")
for name in names:
    g = load_golden(name)
    ln = name.startswith("trainstep_ln_")
    sd = synth.make_state_dict(123, use_layernorm=ln)
    m = GMatcher({"sinkhorn_iterations": int(g["meta"][4]), "pos_loss_weight": float(g["pos_loss_weight"]), "neg_loss_weight": float(g["neg_loss_weight"]),
                  "train_precision": os.environ.get("PROBE_PREC", "bf16x6"), "use_layernorm": ln})
    m.load_state_dict(sd)
    m = m.cuda().train()
    data = train_data(train_pairs(name, g), g, device="cuda")
    m.zero_grad()
    loss, pos, neg = m(data, mode="train")
    loss.backward()
    grads = {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters() if p.grad is not None}
    worst, where, p95 = check_step_gradients(g, grads, rtol=1e9, rtol_p95=1e9)
    print(f"{name:40s} worst {worst:.2e} ({where}) p95 {p95:.2e} loss err {abs(float(loss) - float(g['loss'])):.1e}", flush=True)
