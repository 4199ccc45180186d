#!/usr/bin/env python3
"""GPU probe: time of the training attention calls (forward, backward) for one layer of 2 x n keypoints, per split count."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gims_amd import hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
heads, d = 4, 256
g = torch.Generator().manual_seed(1)
qkv = torch.randn(2 * n, 3 * d, generator=g).cuda()
do = torch.randn(2 * n, d, generator=g).cuda()
for cross in (False, True):
    problems = hip.train_attn_problems([(0, n, n, n), (n, n, 0, n)] if cross else [(0, n, 0, n), (n, n, n, n)])
    for splits in os.environ.get("PROBE_SPLITS", "1,2,4,8").split(","):
        os.environ["GIMS_TRAIN_ATTN_SPLITS"] = splits
        o, lse = hip.train_attention_forward(qkv, problems, heads)
        dq = hip.train_attention_backward(qkv, o, lse, do, problems, heads)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        reps = 20
        ev[0].record()
        for _ in range(reps):
            hip.train_attention_forward(qkv, problems, heads, o, lse)
        ev[1].record()
        for _ in range(reps):
            hip.train_attention_backward(qkv, o, lse, do, problems, heads, dq, precision=hip.PREC_F32)
        ev[2].record()
        for _ in range(reps):
            hip.train_attention_backward(qkv, o, lse, do, problems, heads, dq, precision=hip.PREC_BF16X3)
        ev[3].record()
        torch.cuda.synchronize()
        f, b, b3 = (ev[i].elapsed_time(ev[i + 1]) / reps * 1e3 for i in range(3))
        fl = 2 * heads * n * n * 64 * 2 / 1e12          # one product, both images: TFLOP
        print(f"n={n} cross={int(cross)} splits={splits}: forward {f:7.1f} us ({2 * fl / (f * 1e-6):6.1f} TFLOP/s f32)  backward f32 {b:7.1f} us ({7 * fl / (b * 1e-6):6.1f} TFLOP/s over 7 products)  backward bf16x3 {b3:7.1f} us", flush=True)
