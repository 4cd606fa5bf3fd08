"""Multi-GPU sharding of the sampling job: one process per GPU, independent (audio, seed)
units partitioned by rank, NO collective on the data path -- every unit runs mel -> wave encoder -> DDIM ->
decode -> note grid on its own GPU.  The only communication is one gather of the thresholded note grids
(8 x T bits per chart, packed to bytes) at the end of the job, over torch.distributed (backend "nccl" = RCCL
over xGMI on the GPU box; "gloo" in the CPU tests).

The reference has no counterpart (single process, single device: webui.py:277-482); this is the driver layer
that scripts/mapping.py's `--n_samples` loop (:444-485) becomes when it is spread over a node.
"""
import numpy as np
import torch
import torch.distributed as dist


def partition(n_units, world, rank):
    """Block partition of range(n_units): ranks < n_units % world take one extra unit.  Units that share an
    audio stay adjacent, so a rank can compute that audio's wave features once."""
    base, extra = divmod(int(n_units), int(world))
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def owner_of(unit, n_units, world):
    base, extra = divmod(int(n_units), int(world))
    cut = extra * (base + 1)
    return unit // (base + 1) if unit < cut else extra + (unit - cut) // max(base, 1)


def pack_grids(grid):
    """(n, 8, T) bool -> (n, 8, ceil(T/8)) uint8 (bit j of byte i = cell 8 i + j)."""
    g = np.asarray(grid.detach().cpu() if torch.is_tensor(grid) else grid, dtype=np.uint8)
    return torch.from_numpy(np.packbits(g, axis=-1, bitorder="little"))


def unpack_grids(packed, T):
    g = np.unpackbits(packed.cpu().numpy(), axis=-1, bitorder="little")[..., :T]
    return torch.from_numpy(g.astype(np.bool_))


def gather_grids(local_grid, n_units, group=None, device=None):
    """local_grid: (n_local, 8, T) bool for this rank's partition (in unit order).  Returns the (n_units, 8, T)
    bool tensor on every rank (all_gather of equally padded, bit-packed blocks).  world_size 1 / no process
    group: returns local_grid."""
    if not (dist.is_available() and dist.is_initialized()):
        return local_grid.cpu() if torch.is_tensor(local_grid) else torch.as_tensor(local_grid)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    T = int(local_grid.shape[-1])
    per = -(-int(n_units) // world)                                  # padded block size
    packed = pack_grids(local_grid)
    buf = torch.zeros((per,) + tuple(packed.shape[1:]), dtype=torch.uint8)
    buf[: packed.shape[0]] = packed
    if device is not None:
        buf = buf.to(device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    parts = [out[r][: len(partition(n_units, world, r))] for r in range(world)]
    return unpack_grids(torch.cat(parts, dim=0), T)


def broadcast_tensor(t, src=0, group=None):
    """One-shot broadcast of shared inputs (e.g. the PCM of a song every rank draws seeds for)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(t, src=src, group=group)
    return t
