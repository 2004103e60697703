"""Batch sharding of independent samples across the GPUs of one node.

The reference's only multi-GPU mechanism on the sampling path is ``torch.nn.DataParallel``
(iadb_bn.py:716): weights re-broadcast and activations scattered/gathered on *every* UNet call.
Here each rank (one process per GPU, torch.distributed over RCCL/xGMI) keeps its own weights and
``L``, runs its contiguous chunk of the batch through the whole loop locally, and the only
collective is one gather of the final uint8 images to rank 0.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* if launched by torchrun.
    Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(total: int, rank: int, world: int):
    """Contiguous chunk [begin, begin+count) of `total` samples for `rank`; the first
    total % world ranks take one extra sample."""
    base, rem = divmod(total, world)
    count = base + (1 if rank < rem else 0)
    begin = rank * base + min(rank, rem)
    return begin, count


def gather_images(local_u8: torch.Tensor, counts=None, dst: int = 0):
    """One collective per batch: gather [b_r, H, W, C] uint8 shards to `dst`; returns the
    concatenated [sum b_r, H, W, C] tensor on dst, None elsewhere.  Ragged shards are padded to the
    largest shard (NCCL/RCCL gather needs equal sizes) and trimmed on arrival."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local_u8
    world, rank = dist.get_world_size(), dist.get_rank()
    if counts is None:
        counts = [local_u8.shape[0]] * world
    mx = max(counts)
    buf = local_u8
    if local_u8.shape[0] != mx:
        buf = local_u8.new_zeros((mx,) + tuple(local_u8.shape[1:]))
        buf[: local_u8.shape[0]] = local_u8
    buf = buf.contiguous()
    out = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    return torch.cat([o[:c] for o, c in zip(out, counts)], dim=0)


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value: float, device=None) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
