#!/usr/bin/env python3
"""Times the bf16x6 GEMM (similarity / score products) in isolation, split by diagnostic flags like gemm_probe.py.

usage: python tools/x6_probe.py [n] [problems] [upper]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gims_amd import hip  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    upper = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    g = torch.Generator(device="cpu").manual_seed(0)
    hip.load()
    keep, largs = [], []
    for _ in range(cnt):
        a = torch.randn(n, 256, generator=g).cuda()
        A3 = hip.split_spl3(a)
        W3 = A3 if upper else hip.split_spl3(torch.randn(n, 256, generator=g).cuda())
        out = torch.empty(n, n, dtype=torch.float32, device="cuda")
        largs.append((A3, W3, out))
        keep.append((A3, W3, out))
    buf = torch.empty(256 * cnt, dtype=torch.uint8, device="cuda")
    res = {}
    for name, fl in [("full", 0), ("no epilogue", 0x200), ("epilogue only", 0x100), ("K loop w/o DMA", 0x200 | 0x400), ("K loop w/o MFMA", 0x200 | 0x800), ("full", 0)]:
        args = []
        for A3, W3, out in largs:
            la = hip.linear_args(A3, W3, out=out, precision=hip.PREC_BF16X6, scale=1.0, n=n)
            la.flags = (hip.LINEAR_UPPER if upper else 0) | fl
            args.append(la)
        for _ in range(3):
            hip.linear_batch(args, buf, hip.PREC_BF16X6)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            hip.linear_batch(args, buf, hip.PREC_BF16X6)
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 10 * 1e3
    flops = cnt * 2.0 * n * n * 256 * (0.5 if upper else 1.0)
    print(f"n={n} x{cnt} upper={upper}: " + " | ".join(f"{k} {v:7.1f} us" for k, v in res.items()) +
          f" | MFMA floor (6 passes) {6 * flops / 2.5e15 * 1e6:6.1f} us, output bytes {cnt * n * n * 4 * (0.5 if upper else 1.0) / 1e6:.0f} MB")


if __name__ == "__main__":
    main()
