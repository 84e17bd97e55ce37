"""Data-parallel helpers (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The training step shards by batch rows: samples are independent (no batch statistics; the loss is
a mean over rows), so with equal rows per rank the global-batch gradient is the mean of the
per-rank gradients.  One exchange per step: a SUM all-reduce of the flat gradient buffer, turned
into the mean by the 1/world factor the fused AdamW applies (dg_adamw_step grad_scale).  ln_f has
no gradient (src/model.py:598-599) and lies outside the reduced range on every rank.
These helpers are device-agnostic so the N > 1 logic is testable with gloo on CPU.
"""
from __future__ import annotations

import os
from typing import Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when not launched by it."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend: Optional[str] = None, device: Optional[torch.device] = None):
    rank, local_rank, world = env_world()
    if world == 1:
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if (device is not None and device.type == "cuda") else "gloo"
    if not dist.is_initialized():
        kw = {"device_id": device} if backend == "nccl" and device is not None else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist.group.WORLD


def shard_rows(global_rows: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """rank r takes rows [r*B_local, (r+1)*B_local) of the global batch (offsets or token rows)."""
    n = global_rows.shape[0]
    if n % world:
        raise ValueError(f"global batch {n} is not divisible by world size {world}")
    b = n // world
    return global_rows[rank * b:(rank + 1) * b]


def allreduce_sum_(flat: torch.Tensor, group=None, bucket_elems: int = 0) -> torch.Tensor:
    """in-place SUM all-reduce of a flat buffer, optionally in buckets (async ops joined at the end)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return flat
    if bucket_elems <= 0 or flat.numel() <= bucket_elems:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        return flat
    works = []
    for s in range(0, flat.numel(), bucket_elems):
        works.append(dist.all_reduce(flat[s:s + bucket_elems], op=dist.ReduceOp.SUM, group=group, async_op=True))
    for w in works:
        w.wait()
    return flat


def mean_loss(loss: torch.Tensor, group=None) -> torch.Tensor:
    """global-batch loss for logging = mean of the equal-sized per-rank means."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return loss
    out = loss.detach().clone()
    dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
    return out / dist.get_world_size(group)


def flatten(tensors: Sequence[torch.Tensor]) -> torch.Tensor:
    return torch.cat([t.reshape(-1) for t in tensors])
