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


class GradBucketer:
    """Overlaps the gradient all-reduce with backward: the ONE flat gradient buffer is cut into contiguous buckets of
    ~bucket_bytes (whole parameters; parameters are registered in forward order, so gradients become ready from the LAST
    bucket backwards).  ``mark_ready(i)`` is called from parameter i's post-accumulate-grad hook; when every parameter of a
    bucket has reported, its slice is all-reduced asynchronously (RCCL runs it on its own stream beside the remaining backward
    kernels).  ``finish()`` launches whatever is still pending (parameters that received no gradient this step never report)
    and waits for all handles.  xGMI is point-to-point (7 links x ~153 GB/s per GPU), a ring all-reduce is per-link bound:
    25-50 MB buckets keep each collective bandwidth-bound (>= 0.3 ms) without leaving a long un-overlapped tail (the first
    bucket of the encoder is the last to become ready).  Pure torch.distributed: runs on CPU tensors over gloo in the tests."""

    def __init__(self, flat: torch.Tensor, offsets, numel: int, bucket_bytes: int = 32 << 20, group=None, first_bucket_bytes=None):
        """first_bucket_bytes: size at which bucket 0 -- the FIRST parameters in registration order (the encoder's stem and first
        layers), whose gradients are the LAST to become ready -- is closed; default a quarter of bucket_bytes.  Its all-reduce is
        launched when backward is all but over and is therefore fully exposed: with equal buckets that was 32.6 MB at the benchmark
        model (launched 0.03 ms before the end of backward, profiles/r04_bench_bucketer.json); a quarter-size first bucket exposes
        ~8 MB and lets the bucket behind it (rest of layer 3) start while layers 1 / 2 -- the expensive high-resolution part of the
        encoder's backward -- are still running."""
        self.flat, self.group = flat, group
        if first_bucket_bytes is None:
            first_bucket_bytes = max(1, bucket_bytes // 4)
        n = len(offsets)
        ends = list(offsets[1:]) + [numel]
        self.bucket_of = [0] * n
        self.ranges = []                 # (lo, hi) element ranges, in parameter order
        lo, cur = 0, 0
        es = flat.element_size()
        for i in range(n):
            target = first_bucket_bytes if not self.ranges else bucket_bytes
            # a large parameter behind a half-full bucket starts its own bucket instead of overshooting the target by more than a quarter
            # (spatial_embed.weight at the 40x64 map is 26 MB: behind 26 MB of other parameters it made a 52 MB bucket)
            if offsets[i] > lo and (offsets[i] - lo) * es >= target // 2 and (ends[i] - lo) * es > target + target // 4:
                self.ranges.append((lo, offsets[i]))
                lo = offsets[i]
                target = bucket_bytes
            self.bucket_of[i] = len(self.ranges)
            cur = ends[i]
            if (cur - lo) * es >= target or i == n - 1:
                self.ranges.append((lo, cur))
                lo = cur
        self.members = [0] * len(self.ranges)
        for b in self.bucket_of:
            self.members[b] += 1
        self.reset()

    def reset(self):
        self.pending = list(self.members)
        self.reported = [False] * len(self.bucket_of)      # per parameter: exact detection of a gradient produced twice
        self.next_b = len(self.ranges) - 1      # buckets are launched in strictly DESCENDING order on every rank: the sequence
        self.handles = []                       # of collectives is identical even when ranks' autograd graphs differ (COCO heads)
        self.ready_events = []                  # (bucket, bytes, event on the compute stream at launch) when self.record

    last_ready_events = []
    record = False      # bench.py --force-bucketer: time stamp of every bucket launch on the compute stream (bucket timeline)
    host_launch_ms = 0.0      # (record only) host milliseconds spent inside dist.all_reduce(async_op=True) calls since the last reset_host_ms()
    host_wait_ms = 0.0        # (record only) host milliseconds spent in handle.wait() (stream hand-off back to the compute stream)

    def _launch_ready(self, force: bool = False):
        while self.next_b >= 0 and (force or self.pending[self.next_b] == 0):
            lo, hi = self.ranges[self.next_b]
            if self.record and self.flat.is_cuda:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                self.ready_events.append((self.next_b, (hi - lo) * self.flat.element_size(), ev))
                import time
                t0 = time.perf_counter()
                self.handles.append(dist.all_reduce(self.flat[lo:hi], group=self.group, async_op=True))
                self.host_launch_ms += (time.perf_counter() - t0) * 1e3      # host time inside the collective's launch (autograd thread)
            else:
                self.handles.append(dist.all_reduce(self.flat[lo:hi], group=self.group, async_op=True))
            self.next_b -= 1

    def mark_ready(self, i: int):
        b = self.bucket_of[i]
        if self.reported[i] or b > self.next_b:
            # a second backward before step() (gradient accumulation, two losses): this parameter already reported -- its bucket's
            # all-reduce may be in flight on the RCCL stream while autograd accumulates into the same memory, or (bucket not launched
            # yet) the bucket would be launched one report early; either way the replicas would drift silently
            raise RuntimeError(f"GradBucketer: the gradient of parameter #{i} (bucket {b}, next bucket to launch {self.next_b}, already "
                               f"reported: {self.reported[i]}) was produced twice before FlatAdam.step() (gradient accumulation "
                               "or two backward passes per step); the overlapped bucketed all-reduce supports ONE backward per step -- "
                               "construct FlatAdam(bucket_mb=0) for a single un-overlapped all-reduce in step(); after a backward whose "
                               "step() is skipped call FlatAdam.zero_grad() (it drains and resets the buckets)")
        self.reported[i] = True
        self.pending[b] -= 1
        self._launch_ready()

    def finish(self):
        self._launch_ready(force=True)
        self.drain()

    def drain(self):
        """wait for every collective in flight and start a new round (FlatAdam.zero_grad after a backward whose step() was skipped:
        the flat gradient buffer must not be cleared under an all-reduce, and the next backward must find fresh counters)"""
        if self.record:
            import time
            t0 = time.perf_counter()
        for h in self.handles:
            h.wait()
        if self.record:
            self.host_wait_ms += (time.perf_counter() - t0) * 1e3
        events = self.ready_events
        self.reset()
        self.last_ready_events = events


def union_flags(flags, device, group=None):
    """Element-wise OR across ranks of a list of booleans (which parameters received a gradient this step).  Under the
    reference's DataParallel one optimizer sees the reduce-added gradients, so a parameter is updated when ANY replica used
    it (COCO_Search18: the per-category heads of the categories present anywhere in the global batch); every rank must take
    the same decision or the replicas drift apart.  Only models with conditionally-used parameters need this exchange
    (FlatAdam(conditional_params=True)); it costs one small all-reduce and one host sync per step."""
    if world_size(group) == 1:
        return list(flags)
    t = torch.tensor([1 if f else 0 for f in flags], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return [bool(v) for v in t.tolist()]


def broadcast_module_state_(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """nn.DataParallel replicates replica 0's parameters AND buffers on every forward (AiR/train.py:169-170); with one
    process per GPU the replicas must START identical instead: broadcast every parameter and buffer (BatchNorm running
    statistics, num_batches_tracked) from ``src``.  FlatAdam broadcasts its flat parameter buffer itself; call this for the
    buffers (or after loading a checkpoint on rank 0 only)."""
    if world_size(group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src, group=group)


def assert_replicas_identical(flat: torch.Tensor, group=None) -> None:
    """cheap drift check: the (sum, sum of squares) fingerprint of the flat parameter buffer must agree on all ranks"""
    w = world_size(group)
    if w == 1:
        return
    fp = torch.stack([flat.double().sum(), (flat.double() ** 2).sum()])
    lo, hi = fp.clone(), fp.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    if not torch.equal(lo, hi):
        raise RuntimeError("data-parallel replicas hold different parameters (seed / checkpoint not shared by all ranks?)")


def _slice0(v, lo, hi):
    return v[lo:hi]


def shard_batch(batch: dict, rank: int, world: int) -> dict:
    """DataParallel-style scatter along dim 0 (AiR/train.py:170): every per-sample input splits the same way -- tensors,
    numpy arrays and the python lists the RL batches carry (fix_vectors, performances; AiR/train.py:223-228).  Raises when the
    batch is smaller than the world (an empty shard would launch zero-size kernels)."""
    out = {}
    for k, v in batch.items():
        n = v.shape[0] if hasattr(v, "shape") else len(v)
        if n < world:
            raise ValueError(f"shard_batch: '{k}' has {n} samples for {world} ranks")
        per = (n + world - 1) // world
        lo, hi = rank * per, min(n, (rank + 1) * per)
        if hi <= lo:
            raise ValueError(f"shard_batch: rank {rank} of {world} receives no sample of '{k}' (n = {n})")
        out[k] = _slice0(v, lo, hi)
    return out
