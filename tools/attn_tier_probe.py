#!/usr/bin/env python3
"""A/B of the attention tiers on one box: per-launch time of the attention stage at 2x4096x8 for the three operand formats
(bf16 / IEEE half / split-bf16), with diffuse (default synthetic) and peaked (query / key gain 2.0) weights.

    python tools/attn_tier_probe.py [kpts pairs]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gims_amd import GMatcher, synth  # noqa: E402


def main():
    kpts = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    torch.set_grad_enabled(False)
    inputs = bench.make_inputs(list(range(pairs)), kpts, "cuda:0")
    for wname, gains in (("diffuse", None), ("peaked", {"attn.proj.0": 2.0, "attn.proj.1": 2.0})):
        sd = synth.make_state_dict(123, gains=gains)
        for prec in ("bf16", "f16", "bf16x3"):
            m = GMatcher({"attention_precision": prec}).eval()
            m.load_state_dict(sd)
            for _ in range(3):
                m.match_pairs([dict(d) for d, _ in inputs])
            torch.cuda.synchronize()
            m.enable_timing(True)
            for _ in range(10):
                m.match_pairs([dict(d) for d, _ in inputs])
            torch.cuda.synchronize()
            st = m.stage_times_ms()
            sfx = {"bf16": "", "f16": "_f16", "bf16x3": "_x3"}[prec]
            a = np.sum(st["attn_self" + sfx]) / 10 / 9
            c = np.sum(st["attn_cross" + sfx]) / 10 / 9
            q = np.sum(st["qkv" + sfx]) / 10 / 18
            flops = 2 * pairs * 4 * 4.0 * kpts * kpts * 64
            print(f"{wname:8s} {prec:7s} attention {1e3 * a:7.1f} / {1e3 * c:7.1f} us per self / cross launch ({flops / (c * 1e-3) / 2.5e15:.3f} of the bf16 MFMA peak), "
                  f"qkv {1e3 * q:6.1f} us", flush=True)
            m.enable_timing(False)
            del m


if __name__ == "__main__":
    main()
