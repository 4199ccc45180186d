#!/usr/bin/env python3
"""Golden vectors for ONE TRAINING STEP of the reference, BY RUNNING THE REFERENCE ITSELF (build container only):
``gmodel.train(); loss, pos, neg = gmodel(data, mode='train'); loss.backward()`` exactly as train.py:100, 136-137 does it
(BatchNorm in batch-statistics mode, running statistics updated by the forward pass, gradients by the reference's own
autograd).  `dgl`, `torch_scatter`, `cv2` come from tools/_ref_stubs ("parity unpinned" for the DGL / torch_scatter
arithmetic, see tools/gen_golden.py).

Stored per fixture: the three loss values; for every parameter its gradient -- in full when it has at most 2048 entries,
otherwise a fixed sample of 256 entries plus sum, L2 norm and max |.| --; the BatchNorm running statistics after the step.

    python tools/gen_golden_grads.py
"""
import os
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402
import gen_golden_train as GT  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gims_amd import synth  # noqa: E402

FULL_MAX, SAMPLE = 2048, 256


def sample_index(name: str, numel: int) -> np.ndarray:
    """The entries of a large gradient that are stored: a fixed function of the parameter name (tests recompute it)."""
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    return np.sort(rng.choice(numel, SAMPLE, replace=False)).astype(np.int64)


def one(name, sd, cfg, pairs, rad, pct, ms):
    model = G.RG.GMatcher(dict(cfg))
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    model.train()
    datas = [G.to_data(p, rad, pct, ms) for p in pairs]
    data = {k: (torch.cat([d[k] for d in datas]) if torch.is_tensor(datas[0][k]) else datas[0][k]) for k in datas[0]}
    data["image0"] = np.concatenate([p["image0"] for p in pairs])
    data["image1"] = np.concatenate([p["image1"] for p in pairs])
    matches = np.concatenate([GT.matches_of(b, p["gt_perm"], p["keypoints1"].shape[1]) for b, p in enumerate(pairs)])
    data["matches"] = torch.from_numpy(matches)
    with G.quiet(), torch.enable_grad():
        model.zero_grad()
        loss, pos, neg = model(data, mode="train")
        loss.backward()
    out = dict(loss=np.float64(loss.detach()), pos=np.float64(pos.detach()), neg=np.float64(neg.detach()), matches=matches,
               meta=np.asarray([pairs[0]["keypoints0"].shape[1], rad, pct, ms, cfg.get("sinkhorn_iterations", 100), len(pairs)], dtype=np.int64),
               pos_loss_weight=np.float64(cfg["pos_loss_weight"]), neg_loss_weight=np.float64(cfg["neg_loss_weight"]))
    names = []
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.detach().numpy().astype(np.float32).reshape(-1)
        names.append(k)
        if g.size <= FULL_MAX:
            out["g:" + k] = g
        else:
            out["s:" + k] = g[sample_index(k, g.size)]
            out["n:" + k] = np.asarray([g.astype(np.float64).sum(), np.sqrt((g.astype(np.float64) ** 2).sum()), np.abs(g).max()], dtype=np.float64)
    for k, b in model.named_buffers():
        if k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"):
            out["b:" + k] = b.detach().numpy()
    for s in "01":
        for b in range(len(pairs)):
            out[f"kept{s}_{b}"] = np.asarray(data[f"kept_kpts{s}_indices"][b], dtype=np.int64)
    G.save(name, **out)
    gn = np.sqrt(sum(float((p.grad.double() ** 2).sum()) for p in model.parameters() if p.grad is not None))
    print(f"  {name}: loss {float(loss):.6f} pos {float(pos):.6f} neg {float(neg):.6f}; {len(names)} gradients, global norm {gn:.4e}", flush=True)


def main():
    sd = synth.make_state_dict(123)
    w = dict(GT.WEIGHTS)
    one("trainstep_n256_s1002_i100", sd, {**w}, [synth.make_pair(256, 1002)], 15, 2, 7)
    one("trainstep_n512_s1003_i20", sd, {**w, "sinkhorn_iterations": 20}, [synth.make_pair(512, 1003)], 15, 2, 7)
    # a batch of two with equal kept counts (torch.stack in gmatcher.py:244-249 needs that): batch statistics over B*N positions
    one("trainstep_b2_n64_s1000_i100", sd, {**w}, [synth.make_pair(64, 1000), synth.make_pair(64, 1000, desc_noise=0.2)], 15, 2, 7)
    # sparse canvas: the adaptive graph drops most keypoints (different counts per image) -> most ground-truth rows are remapped
    one("trainstep_n1024sparse_s2001_i20", sd, {**w, "sinkhorn_iterations": 20}, [synth.make_pair(1024, 2001, canvas=(800, 600))], 15, 2, 7)
    # use_layernorm=True (gmatcher.py:19-20, 74-85): the reference's LayerNorm instead of BatchNorm in every MLP
    one("trainstep_ln_n256_s1002_i100", synth.make_state_dict(123, use_layernorm=True), {**w, "use_layernorm": True}, [synth.make_pair(256, 1002)], 15, 2, 7)
    if "--large" in sys.argv:
        # the training configuration of the reference (configs/coco_config.yaml: batch_size 1, train.py:107 max_keypoints 2048)
        one("trainstep_n2048_s1004_i100", sd, {**w}, [synth.make_pair(2048, 1004)], 15, 2, 7)
        # twice the training size (the matcher's headline size); 20 iterations keep the reference's autograd tape within memory here
        one("trainstep_n4096_s1005_i20", sd, {**w, "sinkhorn_iterations": 20}, [synth.make_pair(4096, 1005)], 15, 2, 7)


if __name__ == "__main__":
    main()
