#!/usr/bin/env python3
"""GPU probe: attention layer time at the bench shapes (self layers), TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gims_amd import hip
hip.load()
PRE = os.environ.get('GIMS_ATTN_PRESCALED', '1') != '0'     # the matcher folds the softmax scale into the query projection
for n, imgs in ((4096, 16), (1022, 64)):
    rows = n * imgs
    qkv = torch.randn(rows, 768, device="cuda") * 0.5
    if PRE:
        qkv[:, :256] *= hip.ATTN_Q_SCALE
    qkv = qkv.to(torch.bfloat16)
    pr = torch.tensor([[i * n, n, i * n, n] for i in range(imgs)], dtype=torch.int32, device="cuda")
    osp = torch.empty(rows, 512, dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
        hip.attention(qkv, pr, n, 4, None, 0, 256, 512, out_split=osp, q_prescaled=PRE)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        hip.attention(qkv, pr, n, 4, None, 0, 256, 512, out_split=osp, q_prescaled=PRE)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    fl = 4.0 * n * n * 64 * 4 * imgs
    print(f"n={n} x{imgs} images: {ms*1e3:8.1f} us per layer  {fl/ms*1e-9:7.1f} TFLOP/s  (GIMS_ATTN_EXACT={os.environ.get('GIMS_ATTN_EXACT','0')} q_prescaled={PRE})")
