#!/usr/bin/env python3
"""Host-side profile (cProfile) of the training step: where the Python time of forward + backward goes."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gims_amd import GMatcher, synth  # noqa: E402
from tools.train_bench import batch  # noqa: E402

cfg = {"sinkhorn_iterations": 100, "pos_loss_weight": 0.45, "neg_loss_weight": 1.0}
m = GMatcher(cfg)
m.load_state_dict(synth.make_state_dict(123))
m = m.cuda().train()
ds = [batch(2048, 1000 + i, "cuda") for i in range(8)]


def run(n0, n1):
    for i in range(n0, n1):
        loss, _, _ = m(ds[i], mode="train")
        loss.backward()
        m.zero_grad()
    torch.cuda.synchronize()


run(0, 3)
pr = cProfile.Profile()
pr.enable()
run(3, 8)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
