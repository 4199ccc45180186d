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


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Pair i goes to rank i mod world; every rank gets ceil or floor of n/world pairs."""
    return list(range(rank, n_items, world))


def pair_stats(pair_ids: Sequence[int], outs: Sequence[dict], device) -> torch.Tensor:
    """[n_pairs, 5] float32 record per pair, built on the device without a host sync."""
    rows = []
    for pid, o in zip(pair_ids, outs):
        m0 = o["matches0"].reshape(-1)
        s0 = o["matching_scores0"].reshape(-1)
        valid = m0 >= 0
        nm = valid.sum().to(torch.float32)
        rows.append(torch.stack([torch.tensor(float(pid), device=device),
                                 torch.tensor(float(m0.numel()), device=device),
                                 torch.tensor(float(o["matches1"].numel()), device=device),
                                 nm, (s0 * valid).sum() / nm.clamp(min=1.0)]))
    return torch.stack(rows) if rows else torch.zeros((0, len(STAT_FIELDS)), dtype=torch.float32, device=device)


def gather_stats(stats: torch.Tensor, world: int | None = None) -> torch.Tensor:
    """All-gather the per-pair records of every rank (ragged counts are padded with pair_id = -1 rows and
    dropped again); returns the records of the whole job sorted by pair id, identical on every rank."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return stats[torch.argsort(stats[:, 0])] if stats.numel() else stats
    world = dist.get_world_size() if world is None else world
    n_local = torch.tensor([stats.shape[0]], dtype=torch.int64, device=stats.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local)
    n_max = int(max(int(c.item()) for c in counts))
    padded = torch.full((n_max, stats.shape[1]), -1.0, dtype=stats.dtype, device=stats.device)
    padded[: stats.shape[0]] = stats
    bufs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded)
    allr = torch.cat(bufs)
    allr = allr[allr[:, 0] >= 0]
    return allr[torch.argsort(allr[:, 0])]
