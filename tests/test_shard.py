"""The N > 1 path of the sampling job on CPU: unit partitioning and the end-of-job note-grid
gather, with a real 2-process `gloo` group (the GPU box runs the same code over RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mug import shard


@pytest.mark.parametrize("n,world", [(64, 8), (7, 2), (3, 4), (8, 1), (0, 2), (10, 3)])
def test_partition_covers_every_unit_once(n, world):
    seen = []
    for r in range(world):
        p = shard.partition(n, world, r)
        seen += list(p)
        for u in p:
            assert shard.owner_of(u, n, world) == r
    assert seen == list(range(n))
    sizes = [len(shard.partition(n, world, r)) for r in range(world)]
    assert max(sizes) - min(sizes) <= 1


def test_pack_roundtrip():
    g = torch.from_numpy(np.random.default_rng(0).random((3, 8, 77)) > 0.7)
    assert torch.equal(shard.unpack_grids(shard.pack_grids(g), 77), g)


def _grid_of(unit, T):
    return torch.from_numpy(np.random.default_rng(1000 + unit).random((8, T)) > 0.8)


def _worker(rank, world, port, n_units, T, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = shard.partition(n_units, world, rank)
        local = torch.stack([_grid_of(u, T) for u in mine]) if len(mine) else torch.zeros((0, 8, T), dtype=torch.bool)
        full = shard.gather_grids(local, n_units)
        shared = torch.arange(5, dtype=torch.float32) if rank == 0 else torch.zeros(5)
        shard.broadcast_tensor(shared, src=0)
        q.put((rank, full.numpy(), shared.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_units", [5, 4])
def test_two_rank_gloo_gather(n_units):
    world, T = 2, 45
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_units, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.stack([_grid_of(u, T).numpy() for u in range(n_units)])
    for rank, full, shared in got:
        assert full.shape == want.shape and (full == want).all(), rank
        assert (shared == np.arange(5, dtype=np.float32)).all()
