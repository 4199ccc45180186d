"""Multi-GPU harness of the matcher path: image pairs are independent units (eval_homography.py:161-236 has no
cross-pair state until the final AUC, :237-259), so a list of pairs is partitioned across ranks -- one process
per GPU -- and the ONLY collective is an all-gather of the per-pair match statistics (RCCL over xGMI on the GPU
box: torch.distributed backend "nccl"; "gloo" in the CPU tests).  No data-path collective exists.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.distributed as dist

STAT_FIELDS = ("pair_id", "n_kept0", "n_kept1", "n_matches", "mean_score")
# the record SURVEY 8(e) names for the evaluation all-gather (rank 0 runs pose_auc on the gathered rows)
EVAL_STAT_FIELDS = ("pair_id", "n_kept0", "n_matches", "n_inliers", "err_dlt", "err_ransac", "precision", "recall", "dlt_ok", "ransac_ok")


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Pair i goes to rank i mod world; every rank gets ceil or floor of n/world pairs."""
    return list(range(rank, n_items, world))


def _to_device(values, dtype: str, device) -> torch.Tensor:
    """Small host list -> tensor on `device`.  On the GPU the bytes go through the kernel library's table upload
    (stream-ordered, never blocks the host on the stream); on the CPU (gloo tests) it is a plain tensor."""
    import numpy as np
    arr = np.asarray(values, dtype=dtype)
    if torch.device(device).type == "cuda":
        from . import hip
        return hip.upload(arr, device)
    return torch.from_numpy(arr).to(device)


def pair_stats(pair_ids: Sequence[int], outs: Sequence[dict], device) -> torch.Tensor:
    """[n_pairs, 5] float32 record per pair, built on the device with a handful of batched ops: no host sync and no
    per-pair host->device copies (those would each wait for the stream to drain)."""
    if not outs:
        return torch.zeros((0, len(STAT_FIELDS)), dtype=torch.float32, device=device)
    flat = getattr(outs, "flat", None)
    if flat is not None:            # batch-concatenated outputs of GMatcher.match_pairs (one entry per stream lane)
        flats = flat if isinstance(flat, (list, tuple)) else [flat]
        n0 = [n for f in flats for n in f["n0"]]
        n1 = [n for f in flats for n in f["n1"]]
        m0 = flats[0]["matches0"] if len(flats) == 1 else torch.cat([f["matches0"] for f in flats])
        s0 = flats[0]["scores0"] if len(flats) == 1 else torch.cat([f["scores0"] for f in flats])
    else:
        n0 = [int(o["matches0"].numel()) for o in outs]
        n1 = [int(o["matches1"].numel()) for o in outs]
        m0 = torch.cat([o["matches0"].reshape(-1) for o in outs])
        s0 = torch.cat([o["matching_scores0"].reshape(-1) for o in outs])
    if torch.device(device).type == "cuda" and m0.is_contiguous() and s0.is_contiguous():
        # one table upload + one kernel (gims_pair_stats) instead of ~20 small tensor ops
        import numpy as np
        from . import hip
        offs = np.concatenate([[0], np.cumsum(n0)[:-1]])
        table = hip.upload(np.stack([np.asarray(pair_ids), np.asarray(n0), np.asarray(n1), offs], axis=1).astype(np.int32), device)
        return hip.pair_stats(m0, s0, table)
    # only two tiny (<= a few hundred bytes) host->device copies: larger pageable copies make the HIP runtime pin and
    # unpin the source pages on the fly, which stalls the submitting thread for tens of milliseconds
    host = _to_device([[float(p), float(a), float(b)] for p, a, b in zip(pair_ids, n0, n1)], "float32", device)
    reps = _to_device(n0, "int64", device)
    seg = torch.repeat_interleave(torch.arange(len(outs), device=device), reps, output_size=int(sum(n0)))
    valid = (m0 >= 0).to(torch.float32)
    nm = torch.zeros(len(outs), dtype=torch.float32, device=device).index_add_(0, seg, valid)
    ss = torch.zeros(len(outs), dtype=torch.float32, device=device).index_add_(0, seg, s0 * valid)
    return torch.cat([host, nm[:, None], (ss / nm.clamp(min=1.0))[:, None]], dim=1)


def eval_stats(pair_ids: Sequence[int], datas: Sequence[dict], outs: Sequence[dict], h_gts, device, **eval_kw) -> torch.Tensor:
    """[n_pairs, 10] float32 evaluation record per pair (EVAL_STAT_FIELDS), computed on the device by the kernel library
    (gims_amd.evalh: GT matching from the known homography, precision / recall, 4-point and RANSAC homographies, corner
    errors -- the per-pair body of the reference's eval loop, eval_homography.py:186-226).  No host sync."""
    from . import evalh
    ev = evalh.evaluate_pairs(datas, outs, h_gts, **eval_kw)
    rec = ev["records"]
    col = {k: i for i, k in enumerate(evalh.RECORD_FIELDS)}
    n0 = _to_device([float(o["matches0"].shape[-1]) for o in outs], "float32", device)
    ids = _to_device([float(p) for p in pair_ids], "float32", device)
    pick = [col["n_valid"], col["n_inliers"], col["err_dlt"], col["err_ransac"], col["precision"], col["recall"], col["dlt_ok"], col["ransac_ok"]]
    out = torch.cat([ids[:, None], n0[:, None], rec[:, pick]], dim=1)
    out._eval_keep = ev          # keeps the per-pair outputs (gt0, inlier masks, homographies) alive for the caller
    return out


def eval_summary(gathered: torch.Tensor, min_matches: int = 12) -> dict:
    """AUC@5/10/25 (4-point and RANSAC), mean precision / recall over the gathered evaluation records (host side; what
    rank 0 prints at the end of the reference's eval loop, eval_homography.py:237-259)."""
    from . import evalh
    import numpy as np
    g = gathered.detach().cpu().numpy().astype(np.float64)
    rec = np.zeros((len(g), 16))
    col = {k: i for i, k in enumerate(evalh.RECORD_FIELDS)}
    for name, src in (("n_valid", 2), ("n_inliers", 3), ("err_dlt", 4), ("err_ransac", 5), ("precision", 6), ("recall", 7),
                      ("dlt_ok", 8), ("ransac_ok", 9)):
        rec[:, col[name]] = g[:, src]
    return evalh.summarize(rec, min_matches=min_matches)


def gather_stats(stats: torch.Tensor, world: int | None = None, counts: Sequence[int] | None = None,
                 presorted: bool = False) -> torch.Tensor:
    """All-gather the per-pair records of every rank (ragged counts are padded with pair_id = -1 rows and
    dropped again); returns the records of the whole job sorted by pair id, identical on every rank.
    counts: records per rank when the caller knows them (shard_indices is deterministic) -- then no count exchange and
    no host sync happens here, so the host keeps running ahead of the GPU.  presorted: the caller's records are already
    in ascending pair id (shard_indices yields them so); only used to skip the sort of a single-process run."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        if presorted:        # single process and the caller built the records in ascending pair id: nothing to do
            return stats
        return stats[torch.argsort(stats[:, 0])] if stats.numel() else stats
    world = dist.get_world_size() if world is None else world
    if counts is None:
        n_local = torch.tensor([stats.shape[0]], dtype=torch.int64, device=stats.device)
        cl = [torch.zeros_like(n_local) for _ in range(world)]
        dist.all_gather(cl, n_local)
        counts = [int(c.item()) for c in cl]
    assert len(counts) == world and counts[dist.get_rank()] == stats.shape[0], (counts, stats.shape)
    n_max, n_total = int(max(counts)), int(sum(counts))
    padded = torch.full((n_max, stats.shape[1]), -1.0, dtype=stats.dtype, device=stats.device)
    padded[: stats.shape[0]] = stats
    bufs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded)
    allr = torch.cat(bufs)
    key = torch.where(allr[:, 0] >= 0, allr[:, 0], torch.full_like(allr[:, 0], float("inf")))   # padding rows sort last
    return allr[torch.argsort(key)[:n_total]]
