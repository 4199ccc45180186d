#!/usr/bin/env python3
"""gims_gemm_f32 on the shapes of one training step at 2x2048 keypoints: time, algorithmic TFLOP/s, MFMA occupancy (passes x flops /
2.5 PF) -- run on the GPU box."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gims_amd import hip  # noqa: E402

PREC = hip.PREC_BF16X6 if (len(sys.argv) < 2 or sys.argv[1] == "x6") else hip.PREC_BF16X3
PASSES = 6 if PREC == hip.PREC_BF16X6 else 3
dev = "cuda"
R = 4096          # rows of both images
N = 2048


def t(x):
    return x.t()


def run(name, a, b, out, reps=30, **kw):
    for _ in range(3):
        hip.gemm(a, b, out, precision=PREC, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        hip.gemm(a, b, out, precision=PREC, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    batch = a.shape[0] if a.dim() == 3 else 1
    m, k, n = a.shape[-2], a.shape[-1], b.shape[-2]
    fl = 2.0 * m * n * k * batch
    print(f"{name:44s} m {m:5d} n {n:5d} k {k:5d} b {batch}  {1e6 * dt:7.1f} us  {fl / dt / 1e12:7.1f} TF/s algorithmic  MFMA occupancy {100 * PASSES * fl / dt / 2.5e15:5.1f} %")


x256, x512, x768 = (torch.randn(R, c, device=dev) for c in (256, 512, 768))
w = {(o, i): torch.randn(o, i, device=dev) * 0.05 for o, i in ((768, 256), (256, 256), (512, 512), (256, 512), (512, 256))}
run("forward qkv      X[R,256] W[768,256]^T", x256, w[(768, 256)], torch.empty(R, 768, device=dev))
run("forward mlp0 half X[R,256] W[512,256]^T", x256, w[(512, 256)], torch.empty(R, 512, device=dev))
run("forward mlp3     H[R,512] W[256,512]^T", x512, w[(256, 512)], torch.empty(R, 256, device=dev))
run("input grad       dH[R,512] W[512,512]", x512, t(w[(512, 512)]), torch.empty(R, 512, device=dev))
run("input grad qkv   dQKV[R,768] Wqkv[768,256]", x768, t(w[(768, 256)]), torch.empty(R, 256, device=dev))
run("weight grad      dH^T[512,R] X^T[512,R]", t(x512), t(x512), torch.empty(512, 512, device=dev))
run("weight grad qkv  dQKV^T[768,R] X^T[256,R]", t(x768), t(x256), torch.empty(768, 256, device=dev))
qkv = torch.randn(N, 768, device=dev)
qh = qkv[:, 0:256].view(N, 4, 64).permute(1, 0, 2)
kh = qkv[:, 256:512].view(N, 4, 64).permute(1, 0, 2)
vh = qkv[:, 512:768].view(N, 4, 64).permute(1, 0, 2)
s = torch.empty(4, N, N, device=dev)
o = torch.empty(N, 256, device=dev)
oh = o.view(N, 4, 64).permute(1, 0, 2)
run("S = Q K^T        [4][N,64] x [N,64]^T", qh, kh, s)
run("O = P V          [4][N,N] x V^T[64,N]", s, vh.transpose(1, 2), oh)
run("dV = P^T dO      [4][N,N]^T x dO^T[64,N]", s.transpose(1, 2), oh.transpose(1, 2), vh.clone().transpose(1, 2).transpose(1, 2))
run("dP = dO V^T      [4][N,64] x [N,64]^T", oh, vh, s)
run("dQ = dS K        [4][N,N] x K^T[64,N]", s, kh.transpose(1, 2), oh)
