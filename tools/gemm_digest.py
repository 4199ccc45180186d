#!/usr/bin/env python3
"""Checksums of the big-tile split-bf16 GEMM on MLP-shaped problems (ragged M, two K segments, ReLU / residual / split outputs).
Run before and after a change of the GEMM main loop that must not change results: the digests must be equal."""
import hashlib
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gims_amd import hip  # noqa: E402

g = torch.Generator(device="cpu").manual_seed(5)
dev = "cuda"
for rows in (65536, 50001):
    x = torch.randn(rows, 256, generator=g).to(dev)
    msg = torch.randn(rows, 256, generator=g).to(dev)
    w0 = (torch.randn(512, 512, generator=g) / 22).to(dev)
    w1 = (torch.randn(256, 512, generator=g) / 22).to(dev)
    b0, b1 = torch.randn(512, generator=g).to(dev), torch.randn(256, generator=g).to(dev)
    xs, ms, w0s, w1s = (hip.split_spl32(t) for t in (x, msg, w0, w1))
    hs = torch.zeros(rows, 1024, dtype=torch.bfloat16, device=dev)
    a = hip.linear_args(xs, w0s, a1=ms, bias=b0, act=hip.ACT_RELU, out_split=hs, precision=hip.PREC_BF16X3, spl=True)
    hip._check(hip.load().gims_linear(hip.C.byref(a), hip._stream()), "mlp0")
    out = x.clone()
    os_ = torch.zeros(rows, 512, dtype=torch.bfloat16, device=dev)
    a = hip.linear_args(hs, w1s, bias=b1, residual=out, out=out, out_split=os_, precision=hip.PREC_BF16X3, spl=True)
    hip._check(hip.load().gims_linear(hip.C.byref(a), hip._stream()), "mlp1")
    torch.cuda.synchronize()
    hi, lo = hip.spl32_planes(hs)
    ref = torch.relu(torch.cat([x, msg], 1).double() @ w0.double().T + b0.double())
    err0 = ((hi.double() + lo.double()) - ref).abs().max().item()
    ref1 = x.double() + ref @ w1.double().T + b1.double()
    err1 = (out.double() - ref1).abs().max().item()
    dig = hashlib.sha256(hs.cpu().view(torch.int16).numpy().tobytes() + out.cpu().numpy().tobytes() + os_.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:16]
    print(f"rows={rows} err mlp0 {err0:.2e} mlp1 {err1:.2e} digest {dig}")
