#!/usr/bin/env python3
"""Does the TIME of the half attention kernel depend on the DATA?  Same launch (8 x 2 x 4096 keys, 4 heads), four operand sets:
bf16 kernel on bf16 data; half kernel on full-mantissa half data; half kernel on half data whose mantissas are truncated to bf16's 7 bits;
half kernel on zeros.  (DVFS: the chip clocks to its power budget -- MI355X_MICROARCH.md 'DVFS give-back'.)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gims_amd import hip  # noqa: E402


def main():
    n, pairs = 4096, 8
    rows = 2 * n * pairs
    g = torch.Generator().manual_seed(1)
    x = torch.randn(rows, 768, generator=g)
    x[:, :256] *= hip.ATTN_Q_SCALE
    probs = []
    for p in range(pairs):
        o = 2 * n * p
        probs += [(o, n, o + n, n), (o + n, n, o, n)]
    pr = torch.tensor(probs, dtype=torch.int32, device="cuda")
    out = torch.empty((rows, 512), dtype=torch.bfloat16, device="cuda")
    trunc = x.to(torch.bfloat16).float()
    cases = {"bf16 kernel, bf16 data": (x.to(torch.bfloat16), False),
             "half kernel, half data (11-bit significands)": (x.to(torch.float16).view(torch.bfloat16), True),
             "half kernel, data truncated to bf16's 8-bit significands": (trunc.to(torch.float16).view(torch.bfloat16), True),
             "half kernel, zeros": (torch.zeros(rows, 768, dtype=torch.bfloat16), True),
             "bf16 kernel, zeros": (torch.zeros(rows, 768, dtype=torch.bfloat16), False)}
    for name, (t, f16) in cases.items():
        q = t.cuda()
        for _ in range(5):
            hip.attention(q, pr, n, 4, None, out_split=out, q_prescaled=True, f16=f16)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            hip.attention(q, pr, n, 4, None, out_split=out, q_prescaled=True, f16=f16)
        e1.record()
        torch.cuda.synchronize()
        print(f"{name:60s} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us per launch", flush=True)


if __name__ == "__main__":
    main()
