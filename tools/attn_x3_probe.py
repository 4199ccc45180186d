#!/usr/bin/env python3
"""GPU probe: the split-bf16 attention kernels (GIMS_ATTN_X3) at the bench shapes -- the 32-query-per-wave kernel and the wide one
(GIMS_ATTN_X3W=0 / 2): time per layer, and that the two return the same bits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gims_amd import hip
hip.load()
for n, imgs in ((4096, 16), (2048, 32), (1500, 40), (1022, 64), (700, 8)):
    rows = n * imgs
    x = torch.randn(rows, 768, device="cuda") * 0.7
    x[:, :256] *= 3.0                                  # sharper logits
    spl = hip.split_spl32(x)
    pr = torch.tensor([[i * n, n, i * n, n] for i in range(imgs)], dtype=torch.int32, device="cuda")
    res = {}
    for wide in ("0", "2"):
        os.environ["GIMS_ATTN_X3W"] = wide
        osp = torch.zeros(rows, 512, dtype=torch.bfloat16, device="cuda")
        for _ in range(2):
            hip.attention(spl, pr, n, 4, None, 0, 256, 512, out_split=osp, x3=True)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            hip.attention(spl, pr, n, 4, None, 0, 256, 512, out_split=osp, x3=True)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 5
        fl = 4.0 * n * n * 64 * 4 * imgs
        res[wide] = osp.clone()
        print(f"n={n} x{imgs} images, GIMS_ATTN_X3W={wide}: {ms*1e3:8.1f} us per layer  {fl/ms*1e-9:7.1f} TFLOP/s algorithmic", flush=True)
    print("   outputs bit-identical:", torch.equal(res["0"].view(torch.int16), res["2"].view(torch.int16)))
