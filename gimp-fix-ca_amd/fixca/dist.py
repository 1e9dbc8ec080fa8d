"""Multi-GPU layer for batches of independent images.

The path shards trivially: image i of a batch belongs to rank i mod N.  The only
exchange is one broadcast of the 80-byte parameter block from rank 0 (RCCL over
xGMI on GPUs, gloo in the CPU tests) and one MIN all-reduce of a status word so
every rank learns whether the whole batch succeeded.  No pixel ever crosses GPUs.
"""
import ctypes

import torch
import torch.distributed as dist

from . import FixCaParams


def shard_indices(n_images: int, rank: int, world: int):
    """Indices of the images this rank owns: i mod world == rank."""
    return list(range(rank, n_images, world))


def params_to_tensor(params: FixCaParams, device="cpu") -> torch.Tensor:
    raw = bytes(memoryview(params).cast("B"))
    return torch.tensor(list(raw), dtype=torch.uint8, device=device)


def tensor_to_params(t: torch.Tensor) -> FixCaParams:
    raw = bytes(t.cpu().tolist())
    assert len(raw) == ctypes.sizeof(FixCaParams)
    return FixCaParams.from_buffer_copy(raw)


def broadcast_params(params, src: int = 0, device="cpu") -> FixCaParams:
    """Rank `src` supplies the block, every rank returns the same block.
    `params` may be None on the other ranks."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return params
    if dist.get_rank() == src:
        t = params_to_tensor(params, device)
    else:
        t = torch.zeros(ctypes.sizeof(FixCaParams), dtype=torch.uint8, device=device)
    dist.broadcast(t, src=src)
    return tensor_to_params(t)


def all_ranks_ok(ok: bool, device="cpu") -> bool:
    """True only if every rank reports success."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())
