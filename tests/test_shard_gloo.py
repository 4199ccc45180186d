"""The N > 1 path on CPU: world_size-2 gloo processes exercise the pair sharding and the all-gather of match
statistics (the path's only collective) exactly as bench.py / an eval harness use them on RCCL."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gims_amd import shard


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_pairs, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = shard.shard_indices(n_pairs, rank, world)
        outs = []
        for pid in mine:            # fake per-pair results with a recognisable signature
            n0, n1 = 10 + pid, 20 + pid
            m0 = torch.full((1, n0), -1, dtype=torch.int64)
            m0[0, : pid + 1] = torch.arange(pid + 1)
            outs.append({"matches0": m0, "matches1": torch.full((1, n1), -1, dtype=torch.int64),
                         "matching_scores0": torch.full((1, n0), 0.5)})
        st = shard.pair_stats(mine, outs, "cpu")
        allr = shard.gather_stats(st)
        # the sync-free form (counts known from the deterministic sharding) must return the same table
        counts = [len(shard.shard_indices(n_pairs, r, world)) for r in range(world)]
        assert torch.equal(allr, shard.gather_stats(st, counts=counts))
        q.put((rank, mine, allr.numpy().tolist()))
    finally:
        dist.destroy_process_group()


def test_world2_gloo_shard_and_gather():
    world, n_pairs = 2, 7          # ragged: rank 0 gets 4 pairs, rank 1 gets 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_pairs, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5]
    assert res[0][2] == res[1][2]                      # every rank ends with the same complete table
    table = res[0][2]
    assert [int(r[0]) for r in table] == list(range(n_pairs))
    for r in table:
        pid = int(r[0])
        assert (int(r[1]), int(r[2]), int(r[3])) == (10 + pid, 20 + pid, pid + 1)
        assert abs(r[4] - 0.5) < 1e-6


def test_single_process_passthrough():
    st = torch.tensor([[3.0, 1, 1, 1, 0.5], [1.0, 2, 2, 2, 0.25]])
    out = shard.gather_stats(st)
    assert out[:, 0].tolist() == [1.0, 3.0]
    assert shard.shard_indices(5, 1, 2) == [1, 3]
