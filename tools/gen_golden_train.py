#!/usr/bin/env python3
"""Golden vectors for the FORWARD value of the reference's training loss, BY RUNNING THE REFERENCE ITSELF
(GMatcher.forward(..., mode='train') -> forward_train, models/gmatcher.py:309-386; build container only).

The reference module is put in eval() mode (BatchNorm running statistics: the mode the HIP path implements) and called
exactly like train.py:128-136 does: data['matches'] rows (b, i0, i1) with -1 for "no partner", built from the synthetic
pair's planted correspondences the way train.py:113-125 builds them from torch_find_matches (matches, then image-0 points
without partner, then image-1 points without partner).  `torch_scatter.scatter_mean` is the stub of tools/_ref_stubs
(torch_scatter is not installed; "parity unpinned" for that one call -- its documented semantics are restated there).

    python tools/gen_golden_train.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gims_amd import synth  # noqa: E402

WEIGHTS = {"pos_loss_weight": 0.45, "neg_loss_weight": 1.0}      # configs/coco_config.yaml:25-26


def matches_of(b, gt_perm, n1):
    """train.py:113-125 on one pair: (b, ma_0, ma_1) ++ (b, miss_0, -1) ++ (b, -1, miss_1)."""
    i = np.arange(len(gt_perm))
    pos = gt_perm >= 0
    miss1 = np.setdiff1d(np.arange(n1), gt_perm[pos])
    rows = [np.stack([i[pos], gt_perm[pos]], 1), np.stack([i[~pos], -np.ones((~pos).sum(), np.int64)], 1),
            np.stack([-np.ones(len(miss1), np.int64), miss1], 1)]
    m = np.concatenate(rows).astype(np.int64)
    return np.concatenate([np.full((len(m), 1), b, np.int64), m], 1)


def one(name, model, pairs, rad, pct, ms, iters):
    datas = [G.to_data(p, rad, pct, ms) for p in pairs]
    data = {k: (torch.cat([d[k] for d in datas]) if torch.is_tensor(datas[0][k]) else datas[0][k]) for k in datas[0]}
    data["image0"] = np.concatenate([p["image0"] for p in pairs])
    data["image1"] = np.concatenate([p["image1"] for p in pairs])
    matches = np.concatenate([matches_of(b, p["gt_perm"], p["keypoints1"].shape[1]) for b, p in enumerate(pairs)])
    data["matches"] = torch.from_numpy(matches)
    # the gradient of the loss w.r.t. the score matrix and bin_score, by the reference's own autograd through its unrolled
    # Sinkhorn iterations: log_optimal_transport's input is tapped and keeps its gradient
    cap = {}
    orig = G.RG.log_optimal_transport

    def spy(scores, alpha, iters):
        scores.retain_grad()
        cap["scores"] = scores
        return orig(scores, alpha, iters)
    G.RG.log_optimal_transport = spy
    try:
        with G.quiet(), torch.enable_grad():
            model.zero_grad()
            loss, pos, neg = model(data, mode="train")
            loss.backward()
    finally:
        G.RG.log_optimal_transport = orig
    loss, pos, neg = loss.detach(), pos.detach(), neg.detach()
    dsc = cap["scores"].grad.numpy()                     # (B, n_kept0, n_kept1)
    dbin = float(model.bin_score.grad)
    grads = {"dbin_score": np.float64(dbin)}
    for b in range(len(pairs)):
        d = dsc[b]
        grads[f"dscores_rowsum_{b}"], grads[f"dscores_colsum_{b}"] = d.sum(1).astype(np.float64), d.sum(0).astype(np.float64)
        grads[f"dscores_absmax_{b}"] = np.float64(np.abs(d).max())
        if d.size <= 400 * 400:
            grads[f"dscores_{b}"] = d.astype(np.float32)
        else:                                             # large pairs: a fixed sample of cells instead of the dense matrix
            idx = np.random.default_rng(b).choice(d.size, 8192, replace=False)
            grads[f"dscores_sample_idx_{b}"], grads[f"dscores_sample_{b}"] = idx.astype(np.int64), d.reshape(-1)[idx].astype(np.float32)
    kept = {f"kept{s}_{b}": np.asarray(data[f"kept_kpts{s}_indices"][b], dtype=np.int64) for s in "01" for b in range(len(pairs))}
    G.save(name, matches=matches, loss=np.float64(loss), pos=np.float64(pos), neg=np.float64(neg),
           meta=np.asarray([pairs[0]["keypoints0"].shape[1], rad, pct, ms, iters, len(pairs)], dtype=np.int64),
           pos_loss_weight=np.float64(WEIGHTS["pos_loss_weight"]), neg_loss_weight=np.float64(WEIGHTS["neg_loss_weight"]), **kept, **grads)
    print(f"  loss {float(loss):.6f} pos {float(pos):.6f} neg {float(neg):.6f}; kept {[len(v) for v in kept.values()]}", flush=True)


def main():
    sd = synth.make_state_dict(123)
    m100 = G.ref_model(sd, {**WEIGHTS})
    m20 = G.ref_model(sd, {**WEIGHTS, "sinkhorn_iterations": 20})
    one("trainloss_n256_s1002_i100", m100, [synth.make_pair(256, 1002)], 15, 2, 7, 100)
    one("trainloss_n1024_s1000_i100", m100, [synth.make_pair(1024, 1000)], 15, 2, 7, 100)
    # sparse canvas: the adaptive graph drops most keypoints -> most ground-truth rows are remapped to (b, -1, -1)
    one("trainloss_n1024sparse_s2001_i20", m20, [synth.make_pair(1024, 2001, canvas=(800, 600))], 15, 2, 7, 20)
    # batch of two (equal kept counts, as torch.stack in gmatcher.py:244-249 requires): scatter_mean per batch element
    one("trainloss_b2_n64_s1000_i100", m100, [synth.make_pair(64, 1000), synth.make_pair(64, 1000, desc_noise=0.2)], 15, 2, 7, 100)


if __name__ == "__main__":
    main()
