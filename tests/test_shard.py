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


_RCCL_WORLD1 = r"""
import os, sys, torch, numpy as np
import torch.distributed as dist
sys.path.insert(0, %r)
from mug import shard, train
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
dist.barrier()
g = torch.from_numpy(np.random.default_rng(0).random((4, 8, 4096)) > 0.8)
full = shard.gather_grids(g, 4, device=dev)                      # bit-packed all_gather on the device buffers (mug/job.py's only collective)
assert torch.equal(full, g)
t = torch.arange(5, dtype=torch.float32, device=dev)
shard.broadcast_tensor(t)
red = train.BucketedAllReduce(bucket_bytes=1 << 16, even_single=True)     # the training step's gradient reduction: async all_reduce per bucket
ts = [torch.full((n,), float(i + 1), device=dev) for i, n in enumerate((70000, 3, 20000, 50000))]
for x in ts: red.push(x)
red.finish()
assert red.n_buckets >= 2 and all(float(x[0]) == i + 1 and float(x[-1]) == i + 1 for i, x in enumerate(ts))
lens = [torch.empty(3, dtype=torch.int64, device=dev)]
dist.all_gather(lens, torch.tensor([1, 2, 3], device=dev))
assert lens[0].tolist() == [1, 2, 3]
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_WORLD1_OK", torch.cuda.get_device_name(0))
"""


@pytest.mark.gpu
def test_rccl_path_on_one_gpu():
    """The `nccl` (= RCCL) branches of the job and of the training step on the ONE GPU a test box has: init with a device id, barrier,
    the bit-packed note-grid all_gather on device buffers, the bucketed asynchronous gradient all_reduce, teardown -- in a subprocess so
    that the test session's own process never owns a process group."""
    import subprocess
    import sys
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mug-diffusion_amd")
    r = subprocess.run([sys.executable, "-c", _RCCL_WORLD1 % pkg], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    print(r.stdout[-3000:])
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout
