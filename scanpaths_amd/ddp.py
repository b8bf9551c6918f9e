"""Data-parallel glue: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI on ROCm).

The reference replicates with single-process nn.DataParallel (AiR/train.py:169-170): inputs scattered along dim 0,
outputs gathered, ONE loss over the gathered batch (so it is normalised by the GLOBAL mask sums, :190-197), gradients
reduce-added to GPU 0, BatchNorm statistics per replica.  The MI355X form keeps those semantics with two exchanges per
step and no parameter broadcast:
  1. all-reduce of 2 scalars (the mask sums)  -> every rank scales its local loss by the global normaliser;
  2. all-reduce of the ONE flat gradient buffer (FlatAdam.flat_g), averaged inside the fused Adam kernel.
BatchNorm uses per-replica batch statistics (no SyncBN), like DataParallel.
These helpers contain no device arithmetic, so they run unchanged on CPU tensors with the gloo backend (tests).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def world_size(group=None) -> int:
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def global_mask_normaliser(local_sums: torch.Tensor, group=None) -> torch.Tensor:
    """local_sums = [sum(action_masks), sum(duration_masks)] of this rank's shard.
    Returns S_global / world: with loss_r = numerator_r / (S_global/world), the rank-AVERAGED gradient equals the
    gradient of the reference's single loss  sum_r numerator_r / S_global."""
    w = world_size(group)
    if w == 1:
        return local_sums
    s = local_sums.clone()
    dist.all_reduce(s, group=group)
    return s / w


def allreduce_sum_(flat: torch.Tensor, group=None) -> int:
    """In-place sum all-reduce of the flat gradient buffer; returns the world size (the caller divides)."""
    w = world_size(group)
    if w > 1:
        dist.all_reduce(flat, group=group)
    return w


def union_flags(flags, device, group=None):
    """Element-wise OR across ranks of a list of booleans (which parameters received a gradient this step).  Under the
    reference's DataParallel one optimizer sees the reduce-added gradients, so a parameter is updated when ANY replica used
    it (COCO_Search18: the per-category heads of the categories present anywhere in the global batch); every rank must take
    the same decision or the replicas drift apart."""
    if world_size(group) == 1:
        return list(flags)
    t = torch.tensor([1 if f else 0 for f in flags], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return [bool(v) for v in t.tolist()]


def shard_batch(batch: dict, rank: int, world: int) -> dict:
    """DataParallel-style scatter along dim 0 (AiR/train.py:170): every tensor input splits the same way."""
    out = {}
    for k, v in batch.items():
        n = v.shape[0]
        per = (n + world - 1) // world
        out[k] = v[rank * per:min(n, (rank + 1) * per)]
    return out
