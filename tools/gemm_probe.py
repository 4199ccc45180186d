#!/usr/bin/env python3
"""Times the three per-layer split-bf16 GEMMs of the attentional GNN in isolation (HIP events, one stream).

usage: python tools/gemm_probe.py [rows] [reps]      env GIMS_X3P_TILE selects the tile geometry
Diagnostic flag columns: full | main loop only (no epilogue traffic) | epilogue only (no K loop) | main loop without DMA / without MFMAs |
repeats | main loop with neither DMA nor LDS fragment reads (MFMAs + barriers: the matrix floor at the clock the chip sustains) | without the reads only."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gims_amd import hip  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 65408
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    D = 256

    def spl(r, k):
        x = torch.randn(r, k, generator=g).to(dev)
        return hip.split_spl32(x), x

    dpl, desc = spl(rows, D)
    mpl, _ = spl(rows, D)
    hpl, _ = spl(rows, 2 * D)
    w_qkv, _ = spl(3 * D, D)
    w_m0, _ = spl(2 * D, 2 * D)
    w_m1, _ = spl(D, 2 * D)
    b3, b2, b1 = (torch.randn(n, generator=g).to(dev) for n in (3 * D, 2 * D, D))
    qkv = torch.empty(rows, 3 * D, dtype=torch.bfloat16, device=dev)
    out_h = torch.empty_like(hpl)
    out_d = torch.empty_like(dpl)
    cases = {
        "qkv1 K256 N768 ->bf16 hi-only": (lambda: hip.linear_args(dpl, w_qkv, bias=b3, out_bf16=qkv, precision=hip.PREC_BF16X3, spl=True, flags=hip.LINEAR_HI_ONLY),
                                          2.0 * rows * 256 * 768, rows * (256 * 2 + 768 * 2)),
        "qkv  K256 N768 ->bf16": (lambda: hip.linear_args(dpl, w_qkv, bias=b3, out_bf16=qkv, precision=hip.PREC_BF16X3, spl=True),
                                  2.0 * rows * 256 * 768, rows * (256 * 4 + 768 * 2)),
        "mlp0 K512 N512 relu->spl": (lambda: hip.linear_args(dpl, w_m0, a1=mpl, bias=b2, act=hip.ACT_RELU, out_split=out_h,
                                                             precision=hip.PREC_BF16X3, spl=True),
                                     2.0 * rows * 512 * 512, rows * (512 * 4 + 512 * 4)),
        "mlp1 K512 N256 +res->f32+spl": (lambda: hip.linear_args(hpl, w_m1, bias=b1, residual=desc, out=desc, out_split=out_d,
                                                                 precision=hip.PREC_BF16X3, spl=True),
                                         2.0 * rows * 512 * 256, rows * (512 * 4 + 256 * 4 * 2 + 256 * 4)),
    }
    lib = hip.load()
    import ctypes as C
    st = torch.cuda.current_stream().cuda_stream
    print(f"rows={rows} reps={reps} GIMS_X3P_TILE={os.environ.get('GIMS_X3P_TILE', '(default)')}")
    for name, (mk, flops, byts) in cases.items():
        res = []
        variants = [0, 0] + ([0x1000 + d for d in (1, 2, 3, 4, 6, 8)] if os.environ.get("GIMS_PROBE_DELAY") else [])   # repeats of the full kernel (the first timing of a case runs on a cold clock); 0x1000 + d: odd slots start d x 3.4 us late
        for fl in [0, 0x200, 0x100, 0x200 | 0x400, 0x200 | 0x800] + variants + [0x200 | 0x400 | 0x2000, 0x200 | 0x2000]:
            a = mk()
            a.flags |= fl & ~0xff
            a.conv_reserved = fl & 0xff
            for _ in range(10):
                lib.gims_linear(C.byref(a), st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                lib.gims_linear(C.byref(a), st)
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / reps * 1e3)
        us = res[0]
        print(f"{name:30s} full {us:7.1f} us  ({flops / us * 1e-6:6.1f} TF/s alg, {byts / us * 1e-3:6.0f} GB/s alg) | "
              f"main-only {res[1]:7.1f} us | epilogue-only {res[2]:7.1f} us | main w/o DMA {res[3]:7.1f} | main w/o MFMA {res[4]:7.1f} | variants(full) " + " ".join(f"{v:#x}:{r:.1f}" for v, r in zip(variants, res[5:])) + f" | main MFMA + barriers only {res[-2]:.1f}, main w/o LDS reads {res[-1]:.1f}" + f" | MFMA floor {3 * flops / 2.5e15 * 1e6:5.1f} us, HBM floor {byts / 8e12 * 1e6:5.1f} us")


if __name__ == "__main__":
    main()
