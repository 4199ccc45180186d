#!/usr/bin/env python3
"""Training-step throughput of the HIP path: model.train(); loss = model(data, mode='train')[0];
loss.backward(); optimizer.step() on one synthetic pair of 2 x N keypoints (the reference trains with batch_size 1 and
max_keypoints 2048: configs/coco_config.yaml, train.py:107), next to the oracle's CPU training step on the same input.

    python tools/train_bench.py [--keypoints 2048] [--steps 10] [--no-cpu]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gims_amd import GMatcher, synth  # noqa: E402
from gims_amd.optim import Adam as FusedAdam  # noqa: E402
from tools.gen_pairs import matches_of  # noqa: E402


def batch(n, seed, device):
    pair = synth.make_pair(n, seed)
    d = {k: torch.from_numpy(np.asarray(v)).to(device) for k, v in pair.items() if k not in ("gt_perm", "image0", "image1")}
    d["image0"], d["image1"] = pair["image0"], pair["image1"]
    d.update(device=torch.device(device), radius=15, percentile=2, min_size=7)
    d["matches"] = torch.from_numpy(matches_of(0, pair["gt_perm"], pair["keypoints1"].shape[1])).to(device)
    return d


def measure(keypoints=2048, steps=10, warmup=2, precision="bf16x6", with_cpu=True, optimizer="fused"):
    """The JSON block of one measurement (also embedded in bench.py's line as `train_step`)."""
    cfg = {"sinkhorn_iterations": 100, "pos_loss_weight": 0.45, "neg_loss_weight": 1.0, "train_precision": precision}
    sd = synth.make_state_dict(123)
    m = GMatcher(cfg)
    m.load_state_dict(sd)
    m = m.cuda().train()
    opt = (torch.optim.Adam if optimizer == "torch" else FusedAdam)(m.parameters(), lr=1e-4)      # train.py:53
    fw, bw, st, losses = [], [], [], []
    with torch.enable_grad():
        # the timed region is the reference's own "Mtime" (train.py:135-139: t3 = time_synchronized(); forward; backward; optimizer.step();
        # zero_grad; t4 = time_synchronized()): ONE synchronisation per step.  Rounds 3-4 also synchronised behind the forward and the
        # backward to split the time -- which serialises the host side of the reverse pass behind the device side of the forward; the
        # per-phase numbers now come from extra steps behind the timed ones.
        for i in range(warmup + steps):
            d = batch(keypoints, 1000 + i % 4, "cuda")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            loss, pos, neg = m(d, mode="train")
            loss.backward()
            opt.step()
            opt.zero_grad()
            lv = float(loss.detach())                    # (the synchronisation)
            t3 = time.perf_counter()
            if i >= warmup:
                st.append(t3 - t0), losses.append(lv)
        for i in range(4):
            d = batch(keypoints, 1000 + i % 4, "cuda")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            loss, pos, neg = m(d, mode="train")
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            loss.backward()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            opt.step()
            opt.zero_grad()
            torch.cuda.synchronize()
            fw.append(t1 - t0), bw.append(t2 - t1)
    out = {"metric": "training steps/sec at 2x%d keypoints, batch 1" % keypoints, "value": 1.0 / float(np.median(st)), "unit": "steps/s",
           "ms_per_step": 1e3 * float(np.median(st)), "forward_ms": 1e3 * float(np.median(fw)), "backward_ms": 1e3 * float(np.median(bw)),
           "phase_note": "forward_ms / backward_ms: separate steps with a synchronisation behind each phase (their sum exceeds ms_per_step)", "steps": steps, "loss_first_last": [losses[0], losses[-1]],
           "dtype": "split-%s MFMA products, f32 everything else" % precision, "data": "synthetic",
           "config": {"workload": "1 pair/step of 2x%d synthetic keypoints, 18 layers, 100 Sinkhorn iterations, train() mode forward + backward + Adam (%s)" % (keypoints, "gims_amd.optim.Adam, fused" if optimizer != "torch" else "torch.optim.Adam"), "optimizer": optimizer}}
    if with_cpu:
        from oracle import gims_oracle as O
        cores = min(os.cpu_count() or 1, 16)            # more threads than that make torch's CPU autograd crawl on many-core hosts
        torch.set_num_threads(cores)
        d = batch(keypoints, 1000, "cpu")
        t0 = time.perf_counter()
        O.train_step(sd, d, cfg)
        out["cpu_baseline"] = {"value": 1.0 / (time.perf_counter() - t0), "unit": "steps/s", "cores": cores, "kind": "port",
                               "sample": "1 training step (forward + autograd backward, no optimizer) of oracle/gims_oracle.py on the same pair"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--keypoints", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--precision", default="bf16x6", choices=["bf16x6", "bf16x3"])
    ap.add_argument("--optimizer", default="fused", choices=["fused", "torch"], help="gims_amd.optim.Adam (one fused multi-tensor launch sequence) or torch.optim.Adam")
    a = ap.parse_args()
    print(json.dumps(measure(a.keypoints, a.steps, a.warmup, a.precision, not a.no_cpu, a.optimizer)))


if __name__ == "__main__":
    main()
