#!/usr/bin/env python3
"""Benchmark of the GIMS matcher hot path on MI355X:  image-pairs/sec at 2 x N keypoints.

    python bench.py --gpus N --steps K --warmup W [--kpts 1024] [--pairs 16] [--sinkhorn-iters 100]

A "step" is one pass of the whole hot path (adaptive graph build -> GraphSAGE -> keypoint encoder ->
18-layer self/cross attention -> Sinkhorn -> mutual matching) over one ragged batch of `--pairs` synthetic
image pairs per GPU (BASELINE.json configs[1]: 1024-keypoint pairs, 256-d descriptors, 9x(self,cross) GNN
layers, 100 Sinkhorn iterations).  Inputs are resident in HBM before the timed region.  Pairs are
independent units: with N > 1 every rank (one per GPU, RCCL) matches its own pairs and the per-pair match
statistics are all-gathered over xGMI each step -- the only collective of the path (weak scaling).

Rank 0 prints ONE JSON line: whole-job pairs/s, the roofline of the dominant kernel (timed live with HIP
events on the launch stream) and the CPU baseline (the oracle timed on this host's cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0       # HBM3E spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_inputs(pair_ids, kpts, device):
    from gims_amd import synth
    datas = []
    for pid in pair_ids:
        pair = synth.make_pair(kpts, 1000 + pid)
        d = {k: torch.from_numpy(v).to(device) for k, v in pair.items() if k not in ("gt_perm", "image0", "image1")}
        d["image0"], d["image1"] = pair["image0"], pair["image1"]
        d.update(device=torch.device(device), radius=15, percentile=2, min_size=7)
        datas.append((d, pair["gt_perm"]))
    return datas


def cpu_baseline(kpts, iters, budget_s=20.0):
    """The oracle (CPU restatement of the reference, oracle/gims_oracle.py) on this host's cores."""
    from gims_amd import synth
    from oracle import gims_oracle as O
    # intra-op threads: the small per-op tensors of this path stop scaling (and then collapse from
    # oversubscription) well before the host's core count, so the baseline uses min(cores, 16) threads
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    sd = synth.make_state_dict(123)
    done, t_used = 0, 0.0
    while True:
        pair = synth.make_pair(kpts, 1000 + done)
        d = {k: torch.from_numpy(v) for k, v in pair.items() if k not in ("gt_perm", "image0", "image1")}
        d["image0"], d["image1"] = pair["image0"], pair["image1"]
        d.update(device=torch.device("cpu"), radius=15, percentile=2, min_size=7)
        t0 = time.perf_counter()
        with torch.no_grad():
            O.gmatcher_forward(sd, d, {"sinkhorn_iterations": iters})
        t_used += time.perf_counter() - t0
        done += 1
        if t_used > budget_s * 0.75 or done >= 64:
            break
    return {"value": done / t_used, "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{done} pair(s) of 2x{kpts} keypoints, {iters} Sinkhorn iterations, oracle/gims_oracle.py "
                      f"(torch CPU, {t_used:.1f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--kpts", type=int, default=1024)
    ap.add_argument("--pairs", type=int, default=32, help="image pairs per step per GPU")
    ap.add_argument("--sinkhorn-iters", type=int, default=100)
    ap.add_argument("--linear-precision", default="bf16x3", choices=["bf16x3", "f32"])
    ap.add_argument("--streams", type=int, default=1, help="independent sub-batches per step on separate HIP streams (1 = single stream)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # GIMS_BENCH_BACKEND=gloo + fewer GPUs than ranks is a DRY RUN of the multi-rank code path on one GPU (ranks share
        # the device; combine with GIMS_OT_RESIDENT=0: the on-chip Sinkhorn kernel needs the whole GPU to itself)
        backend = os.environ.get("GIMS_BENCH_BACKEND", "nccl")
        local_dev = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_dev)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    dev = f"cuda:{torch.cuda.current_device()}"
    torch.set_grad_enabled(False)

    import __graft_entry__
    if world > 1:            # one rank per node (re)builds a stale library, the others wait: no concurrent hipcc runs on one tree
        if local_rank == 0:
            __graft_entry__.build()
        dist.barrier()
    __graft_entry__.build()
    from gims_amd import GMatcher, shard, synth

    model = GMatcher({"sinkhorn_iterations": args.sinkhorn_iters, "linear_precision": args.linear_precision,
                      "streams": args.streams}).eval()
    model.load_state_dict(synth.make_state_dict(123))
    my_pairs = shard.shard_indices(world * args.pairs, rank, world)     # pair i -> rank i mod world
    rank_counts = [len(shard.shard_indices(world * args.pairs, r, world)) for r in range(world)]
    inputs = make_inputs(my_pairs, args.kpts, dev)
    torch.cuda.synchronize()

    host_t = {"match_pairs": [], "stats": []}

    def step():
        datas = [dict(d) for d, _ in inputs]              # shallow copies: forward mutates the dict, tensors stay resident
        t0 = time.perf_counter()
        outs = model.match_pairs(datas)
        host_t["datas"] = datas
        t1 = time.perf_counter()
        # per-pair match statistics, all-gathered over RCCL/xGMI: the path's only collective
        stats = shard.gather_stats(shard.pair_stats(my_pairs, outs, dev), counts=rank_counts, presorted=my_pairs == sorted(my_pairs))
        ms = torch.cuda.memory_stats()
        host_t.setdefault("dev_alloc", []).append((ms.get("num_device_alloc", 0), ms.get("num_device_free", 0), ms.get("reserved_bytes.all.current", 0) >> 20))
        host_t["match_pairs"].append(1e3 * (t1 - t0))
        host_t["stats"].append(1e3 * (time.perf_counter() - t1))
        return outs, stats

    for _ in range(args.warmup):
        step()
    # long-lived objects (weights, inputs, packed planes) are moved out of the cyclic GC's working set: a full
    # collection over them costs tens of milliseconds and would land inside a timed step
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()           # and no collection at all inside the K timed steps (re-enabled right after them)
    model.enable_timing(os.environ.get("GIMS_BENCH_NO_STAGE_TIMERS") is None)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        outs, stats = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    et = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(et, op=dist.ReduceOp.MAX)
    elapsed = float(et.item())
    if model._timers is None:          # diagnostic mode: no stage timers -> no roofline section
        log("host ms/step  match_pairs: " + " ".join(f"{x:6.2f}" for x in host_t["match_pairs"][-args.steps:]))
        if rank == 0:
            print(json.dumps({"value": world * args.pairs * args.steps / elapsed, "ms_per_step": 1e3 * elapsed / args.steps, "diagnostic": True}))
        return
    stage = model.stage_times_ms()
    stage_host = model.stage_host_ms()
    marks = list((model._timers or {}).get("_host_marks", []))
    log("host ms/step  match_pairs: " + " ".join(f"{x:6.2f}" for x in host_t["match_pairs"][-args.steps:]))
    for row in marks:
        log("host marks (ingest, run, outputs, info-sync wait) ms: " + " ".join(f"{x:7.2f}" for x in row[2]))
    log("device allocs/frees/reserved MiB per step: " + " ".join(str(x) for x in host_t["dev_alloc"][-args.steps:]))
    log("host ms/step  stats      : " + " ".join(f"{x:6.2f}" for x in host_t["stats"][-args.steps:]))
    model.enable_timing(False)

    # ---- correctness guard of what was timed: planted correspondences must be recovered
    assert stats.shape[0] == world * args.pairs, "the all-gather must return one record per pair of the whole job"
    o, (d, gt) = outs[0], inputs[0]
    m0 = o["matches0"][0].cpu().numpy()
    st = stats.cpu().numpy()

    # ---- evaluation of what was timed (outside the timed region): the synthetic pairs are a permutation + jitter of one
    # point set, i.e. their ground-truth homography is the identity; GT matching, precision / recall, 4-point and RANSAC
    # homographies and the corner-error AUC run on the device (gims_eval_pairs) and the per-pair records are all-gathered
    # exactly like the reference's eval loop would (eval_homography.py:186-259)
    eval_rec = shard.gather_stats(shard.eval_stats(my_pairs, host_t["datas"], outs, [np.eye(3, dtype=np.float32)] * len(my_pairs), dev,
                                                   ransac_iters=2000, seed=1), counts=rank_counts)
    eval_summary = shard.eval_summary(eval_rec)

    if rank == 0:
        total_pairs = world * args.pairs * args.steps
        value = total_pairs / elapsed
        flats = outs.flat if isinstance(outs.flat, (list, tuple)) else [outs.flat]
        problems = [(a, b) for f in flats for a, b in zip(f["n0"], f["n1"])]
        nl = getattr(model, "n_lanes_last", 1)          # stream lanes: each stage is launched once per lane
        stage_ms = {k: float(np.sum(v)) / args.steps for k, v in stage.items()}
        for k, v in stage.items():
            per = np.asarray(v).reshape(args.steps, -1).sum(1)
            perh = np.asarray(stage_host[k]).reshape(args.steps, -1).sum(1)
            log(f"stage {k:14s} gpu ms/step: " + " ".join(f"{x:7.2f}" for x in per) + "   | host ms/step: " + " ".join(f"{x:7.2f}" for x in perh))
        # algorithmic work per step on this rank (SURVEY 8d formulas, on the kept counts)
        attn_flops_layer = sum(1024.0 * (a * a + b * b) for a, b in problems)          # self layer (both images)
        cross_flops_layer = sum(1024.0 * (2 * a * b) for a, b in problems)
        n_rows = sum(a + b for a, b in problems)
        fused = bool(model.config["fuse_merge"])
        lpl = 3 if fused else 4                         # linear_x3p launches per layer
        # executed linear flops per layer (with the merge conv folded into MLP0 the 256x256 merge GEMM disappears)
        lin_flops_layer = 2.0 * n_rows * (3 * 256 * 256 + (0 if fused else 256 * 256) + 512 * 512 + 512 * 256)
        ot_bytes = sum(2.0 * args.sinkhorn_iters * (a + 1) * (b + 1) * 4 for a, b in problems)
        n_self = sum(1 for t in model.config["transformer_layers"] if t == "self")
        n_cross = len(model.config["transformer_layers"]) - n_self
        L = n_self + n_cross
        per_step = lambda name: float(np.sum(stage[name])) / args.steps          # noqa: E731
        # kernel -> (bound, algorithmic work per launch, avg launch ms (HIP events on the launch stream), peak, unit, launches/step)
        # The "qkv" and "mlp" stages contain ONLY launches of linear_x3p_kernel (1 and 3 per layer).
        cand = {
            "linear_x3p_kernel": ("mfma", lin_flops_layer / (lpl * nl), (per_step("qkv") + per_step("mlp")) / (lpl * L * nl), PEAK_BF16_TFLOPS, "TFLOP/s", lpl * L * nl),
            "attention_bf16_kernel": ("mfma", (n_self * attn_flops_layer + n_cross * cross_flops_layer) / (L * nl),
                                      (per_step("attn_self") + per_step("attn_cross")) / (L * nl), PEAK_BF16_TFLOPS, "TFLOP/s", L * nl),
        }
        # Sinkhorn: SURVEY 8(d)'s algorithmic bytes (two sweeps of the (N+1)x(M+1) matrix per iteration).  Streamed path:
        # one ot_iter_kernel launch per iteration.  Resident path: `ot_plan` launches per step run ALL iterations with the
        # matrix held on chip -- no HBM traffic in the loop, so the "HBM rate" it is priced at can exceed the 8 TB/s peak.
        ot_plan = int(getattr(model, "sinkhorn_plan_last", 0))
        ot_name = "ot_resident_kernel" if ot_plan > 0 else "ot_iter_kernel"
        ot_launches = ot_plan if ot_plan > 0 else args.sinkhorn_iters
        cand[ot_name] = ("hbm", ot_bytes / max(1, ot_launches), per_step("sinkhorn") / max(1, ot_launches), PEAK_HBM_GBS, "GB/s", ot_launches)
        if args.linear_precision != "bf16x3":
            cand["linear_f32_kernel"] = cand.pop("linear_x3p_kernel")[:3] + (157.3, "TFLOP/s", lpl * L)
        totals = {k: v[2] * v[5] for k, v in cand.items()}
        dom = max(totals, key=totals.get)
        rate = lambda v: v[1] / (v[2] * 1e-3) / (1e12 if v[4] == "TFLOP/s" else 1e9)     # noqa: E731
        bound, work, ms, peak, unit, n_launch = cand[dom]
        traffic = None
        dom_kernel_name = dom
        tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")       # HBM bytes per launch from rocprofv3 --pmc runs of this command
        if os.path.exists(tf):
            tab = json.load(open(tf)).get(f"{args.kpts}x{args.pairs}", {})
            # the attention stage runs attention8_bf16_kernel (8-wave workgroups) on large launches, attention_bf16_kernel else
            t = tab.get("attention8_bf16_kernel") if dom == "attention_bf16_kernel" and "attention8_bf16_kernel" in tab else tab.get(dom)
            traffic = t.get("hbm_bytes_per_launch") if t else None
            if dom == "attention_bf16_kernel" and "attention8_bf16_kernel" in tab:
                dom_kernel_name = "attention8_bf16_kernel"
        if nl > 1:
            dom_note_extra = (f" NOTE: {nl} independent sub-batches run on separate HIP streams, so kernels of different lanes overlap in time; "
                              "per-launch durations (HIP events and rocprofv3 alike) include that time-sharing and read LOWER than on an idle GPU -- "
                              "run with --streams 1 for isolated kernel durations")
        else:
            dom_note_extra = ""
        if dom in ("ot_iter_kernel", "ot_resident_kernel"):
            dom_note = ("`achieved` uses SURVEY 8(d)'s algorithmic bytes (TWO sweeps of the (N+1)x(M+1) matrix per iteration); the streamed "
                        "kernel reads the matrix ONCE per iteration and the resident kernel keeps it on chip for all iterations, so frac "
                        "can exceed 1; real HBM bytes are in `traffic`")
        elif dom == "attention_bf16_kernel":
            dom_note = ("flash-style attention, head dim 64: per 64-key tile a wave issues 16 MFMAs (512 matrix-pipe cycles) against ~165 "
                        "VALU/transcendental issues (660 cycles) for the online softmax, so the softmax, not the matrix pipe, bounds it "
                        "(DESIGN.md 4.2); `achieved` counts 4*N*M*64 flops per head")
        else:
            dom_note = ("linear_x3p_kernel issues 3 bf16 MFMA passes per algorithmic product (split-bf16 hi*hi+hi*lo+lo*hi, f32-class accuracy) in "
                        "the two MLP GEMMs of a layer and 1 pass (plain bf16, GIMS_LINEAR_HI_ONLY) in its Q/K/V projection, whose result is rounded "
                        "to bf16 for the attention kernel anyway; `achieved` counts ALGORITHMIC flops 2MNK averaged over the three launches per "
                        "layer, so its ceiling against the 2.5 PF/s bf16 peak is about 0.4")
        # second view of the GEMM launches: they are short-K products (K = 256 / 512) over 4-byte-per-element operands and
        # results, so per launch they also move a lot of HBM: algorithmic bytes = rows x (Q/K/V 256x2 in + 768x2 out, MLP0
        # 512x4 in + 512x4 out, MLP1 512x4 in + 256x4 residual + 256x4 f32 out + 256x4 split out) / launches per layer
        lin_bytes_launch = n_rows * (256 * 2 + 768 * 2 + 512 * 4 + 512 * 4 + 512 * 4 + 3 * 256 * 4) / (lpl * nl) if fused else None
        roofline = {"kernel": dom_kernel_name, "bound": bound, "achieved": rate(cand[dom]), "peak": peak, "unit": unit,
                    "frac": rate(cand[dom]) / peak, "traffic": traffic,
                    "avg_launch_ms": float(ms), "launches_per_step": n_launch, "algorithmic_work_per_launch": work,
                    "note": dom_note + dom_note_extra,
                    "all": {k: {"bound": v[0], "avg_launch_ms": float(v[2]), "launches_per_step": v[5], "achieved": rate(v), "unit": v[4],
                                "peak": v[3], "frac": rate(v) / v[3]} for k, v in cand.items()}}
        if lin_bytes_launch and "linear_x3p_kernel" in cand:
            lms = cand["linear_x3p_kernel"][2]
            roofline["all"]["linear_x3p_kernel"]["hbm_view"] = {
                "algorithmic_bytes_per_launch": lin_bytes_launch, "achieved": lin_bytes_launch / (lms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": lin_bytes_launch / (lms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                "note": "SURVEY 8(d) prices the linears against the MFMA peak (the `frac` above); with K = 256 / 512 the same launches sit at this "
                        "fraction of the HBM peak as well -- neither roof is reached, prologue / epilogue phases do not overlap (DESIGN.md 4.3)"}
        k0 = host_t["datas"][0]["kept_kpts0_indices"][0].cpu().numpy()
        k1 = host_t["datas"][0]["kept_kpts1_indices"][0].cpu().numpy()
        v = m0 >= 0
        correct = int((k1[m0[v]] == gt[k0[v]]).sum())
        assert v.sum() > 0.5 * args.kpts and correct > 0.9 * v.sum(), ("benchmark output is not a valid matching", int(v.sum()), correct)
        res = {
            "metric": f"image-pairs/sec at 2x{args.kpts} keypoints", "value": value, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16 MFMA attention and Q/K/V projection + split-bf16x3 (f32-class) MFMA linears + bf16x6 (f32-class) similarity and score GEMMs + f32 Sinkhorn" if args.linear_precision == "bf16x3"
                     else "bf16 MFMA attention + f32 MFMA linears + f32 Sinkhorn",
            "data": "synthetic",
            "config": {"workload": f"{args.pairs} pairs/step/GPU of 2x{args.kpts} synthetic keypoints (kept {problems[0][0]}/{problems[0][1]} after AGC r=15 p=2 m=7), "
                                   f"256-d descriptors, 18 attentional layers (9 self + 9 cross), {args.sinkhorn_iters} Sinkhorn iterations, match_threshold 0.2",
                       "pairs_per_step_per_gpu": args.pairs, "keypoints": args.kpts, "sinkhorn_iterations": args.sinkhorn_iters,
                       "parallelism": f"pairs sharded over {world} GPU(s), all-gather of match statistics; {nl} stream lane(s) per GPU"},
            "roofline": roofline,
            "stage_ms_per_step": stage_ms,
            "matches_pair0": {"matched": int(v.sum()), "correct_vs_planted": correct},
            "stats_rows_gathered": int(st.shape[0]), "stat_fields": list(shard.STAT_FIELDS),
            "eval": {"note": "quality of the timed outputs against the planted correspondences (identity homography): GT matching, "
                             "precision / recall, corner-error AUC of the 4-point and RANSAC homographies (gims_eval_pairs + all-gather)",
                     **{k: (round(v, 3) if isinstance(v, float) else [round(x, 3) for x in v] if isinstance(v, list) else v)
                        for k, v in eval_summary.items()}},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.kpts, args.sinkhorn_iters)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
