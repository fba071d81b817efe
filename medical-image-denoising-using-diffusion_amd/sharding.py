"""Data-parallel sharding of a batch of independent images over the GPUs of one node.

The reference is single-device; the path shards trivially because
images never interact: GroupNorm and attention are per sample (DDIMModel.py:116,146), ``t`` is
the same for the whole batch (:275) and the DDIM variant draws no random numbers.  Rank r of W
takes the contiguous block ``[r*B/W, (r+1)*B/W)``, runs every step with zero communication,
and ONE all-gather (RCCL over xGMI when the backend is "nccl") collects the outputs.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(batch: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, equal blocks; the batch must divide evenly (all_gather_into_tensor needs it)."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    if batch % world_size:
        raise ValueError(f"batch {batch} is not divisible by world size {world_size}")
    per = batch // world_size
    return rank * per, (rank + 1) * per


def gather_outputs(local: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """[B/W,C,H,W] on every rank -> [B,C,H,W] on every rank, rank-major (= original order)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    local = local.contiguous()
    full = local.new_empty((world * local.shape[0],) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(full, local, group=group)
    return full


def denoise_sharded(denoise_fn: Callable[[torch.Tensor], torch.Tensor], noisy_full: torch.Tensor,
                    group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Every rank holds the full batch; each denoises its block, one all-gather at the end."""
    if not (dist.is_available() and dist.is_initialized()):
        return denoise_fn(noisy_full)
    lo, hi = shard_bounds(noisy_full.shape[0], dist.get_world_size(group), dist.get_rank(group))
    return gather_outputs(denoise_fn(noisy_full[lo:hi]), group)
