#!/usr/bin/env python3
"""GPU probe: host time per call of the training step's wrappers (tiny problems: the GPU keeps up, what is timed is Python + the C launch path)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gims_amd import hip
import gims_amd.hip as H

a = torch.randn(256, 256, device="cuda"); b = torch.randn(256, 256, device="cuda"); out = torch.empty(256, 256, device="cuda")
bias = torch.randn(256, device="cuda")
big = torch.randn(1024, 256, device="cuda")
def t(f, n=3000):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return dt
print("gemm(a, b, out, bias)            %.2f us" % t(lambda: H.gemm(a, b, out, bias=bias)))
print("gemm(a.t(), b.t()) new out       %.2f us" % t(lambda: H.gemm(a.t(), b.t())))
print("gemm k=1024 (split-K + fold)     %.2f us" % t(lambda: H.gemm(big.t(), big.t())))
print("colsum                           %.2f us" % t(lambda: H.colsum(a)))
print("_operand                         %.2f us" % t(lambda: H._operand(a)))
print("Gemm struct                      %.2f us" % t(lambda: H.Gemm(1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,0,1.0,0.0)))
print("torch.empty                      %.2f us" % t(lambda: torch.empty((256,256), device='cuda')))
print("a.t()                            %.2f us" % t(lambda: a.t()))
print("view chain                       %.2f us" % t(lambda: a[10:100, 0:64].view(90, 4, 16).permute(1,0,2)))
ev = torch.cuda.Event()
print("event record + wait_event        %.2f us" % t(lambda: (ev.record(), torch.cuda.current_stream().wait_event(ev))))
print("torch.cuda.Event()               %.2f us" % t(lambda: torch.cuda.Event()))
sg = hip.segments([(0, 128), (128, 128)])
g = torch.ones(256, device="cuda"); rm = torch.zeros(256, device="cuda"); rv = torch.ones(256, device="cuda")
print("batchnorm_train_forward          %.2f us" % t(lambda: H.batchnorm_train_forward(a, sg, g, g, 1e-5, 0.1, rm, rv, True)))
