"""Fused clip_grad_norm_ + Adam over ONE flat fp32 buffer (AiR/train.py:116-117,200-202).

MI355X-first layout: all parameters of the model are re-pointed to views of a single contiguous HBM
buffer (and their .grad to views of a second one), so
  * zero_grad is one memset,
  * the global gradient norm is one reduction and clip + Adam one elementwise kernel (2 launches instead
    of ~1100 for 376 tensors),
  * data parallelism all-reduces ONE buffer (a few large RCCL calls over xGMI instead of 376 small ones).
Views keep each parameter's own strides (conv weights stay channels_last = [Co][KH][KW][Ci]), every
offset is padded to 16 bytes for the float4 kernels, and padding stays exactly zero.

Semantics = torch.nn.utils.clip_grad_norm_(params, clip) followed by torch.optim.Adam(lr, betas, eps,
weight_decay) (L2 folded into the gradient; not AdamW).  state_dict()/load_state_dict() use torch.optim.Adam's
format so reference checkpoints ({"model":..., "optimizer":...}, utils/checkpointing.py:93-110) round-trip.
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch

from . import hip
from .hip import check, ptr

ALIGN = 4  # elements (16 bytes)


class _FlatSlot:
    """Marker FlatAdam leaves on a parameter (``p._sp_flat``): lets a backward kernel write the parameter's gradient straight into its
    view of the flat gradient buffer instead of handing autograd a temporary to add (functional._take_grad_view).  Only the FIRST
    gradient since zero_grad() may do so (the view then holds zeros); later ones take autograd's accumulate path."""
    __slots__ = ("opt", "i")

    def __init__(self, opt, i):
        self.opt, self.i = opt, i

    def peek(self, p):
        o = self.opt
        g = p.grad
        if not o._fresh[self.i] or g is None or g.data_ptr() != o.flat_g.data_ptr() + 4 * o._offs[self.i]:
            return None
        return g

    def take(self):
        self.opt._fresh[self.i] = False

    def done(self):
        # The parameter's AccumulateGrad node still runs once every contribution of this backward has been produced -- with an undefined
        # gradient when this was the only one -- and this torch fires the post-accumulate hook then (gradient-ready bookkeeping, bucketed
        # all-reduce): reporting here as well made every such parameter report twice.  Should a torch version skip the hook for an
        # undefined gradient, FlatAdam.step() reports the parameters still pending.
        self.opt._direct_pending[self.i] = True


class FlatAdam(torch.optim.Optimizer):
    """clip: max total gradient norm (0 = no clipping).  Data parallel (torch.distributed initialised, process_group not
    False): the flat parameter buffer is broadcast from rank 0 at construction (nn.DataParallel replicates replica 0's weights,
    AiR/train.py:169-170 -- replicas that start different would silently stay different), the flat gradient buffer is
    all-reduced in ~bucket_mb buckets launched from the post-accumulate-grad hooks so the exchange overlaps backward
    (ddp.GradBucketer), and the average is folded into the Adam kernel.
    conditional_params: the model has parameters that only some batches use (COCO_Search18 per-category heads).  True: the
    "received a gradient" flags are OR-ed across ranks every step (one small all-reduce + host sync).  False (AiR / OSIE):
    no exchange; a parameter without a gradient under data parallelism then raises instead of letting the replicas drift.
    Which parameters are stepped -- torch.optim.Adam skips ``p.grad is None``:
      * ``zero_grad()`` / ``zero_grad(set_to_none=True)`` (torch >= 2.0 default, the torch installed here): a parameter that
        received no gradient in THIS backward is skipped (no decay, no moment update, no step increment);
      * ``zero_grad(set_to_none=False)`` (the only behaviour of the reference's pinned torch==1.6.0, sp_baseline.yml:116):
        gradients are zeroed, not dropped, so a parameter that has EVER received a gradient keeps being decayed and
        momentum-stepped with a zero gradient (a COCO head whose category is absent from later batches).
    After a backward pass that ABORTS (an exception in a hook, out of memory, a non-finite-loss "skip this batch"): call
    ``zero_grad()`` before the next backward -- it drains the bucketed all-reduces in flight and resets the bookkeeping of the gradients
    that were written straight into the flat buffer (functional._take_grad_view); a parameter with a user tensor hook
    (``p.register_hook``) always takes autograd's ordinary path, so the hook sees its gradient."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 clip: float = 0.0, process_group=None, conditional_params: bool = False, bucket_mb: float = 32.0,
                 reference_zero_grad: Optional[bool] = None, force_bucketer: bool = False):
        params = [p for p in params if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, clip=clip))
        assert len(self.param_groups) == 1
        dev = params[0].device
        if dev.type != "cuda":
            raise hip.HipError("FlatAdam needs parameters on a HIP device (no CPU path)")
        offs, total = [], 0
        for p in params:
            assert p.dtype == torch.float32 and p.device == dev
            offs.append(total)
            total += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self._params, self._offs, self.numel = params, offs, total
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        self._sumsq = torch.zeros(1, dtype=torch.float64, device=dev)
        self._steps = [0] * len(params)          # per-parameter Adam step (torch.optim.Adam keeps it per parameter)
        self._touched = [False] * len(params)    # did this backward produce a gradient for the parameter?
        self._fresh = [True] * len(params)       # nothing has been written into the parameter's gradient view since it was zeroed
        self._direct_pending = [False] * len(params)      # gradient written in place, post-accumulate hook not seen yet (_FlatSlot.done)
        self._sticky = False                     # zero_grad(set_to_none=False) semantics, see the class docstring
        # reference_zero_grad: a bare zero_grad() zero-fills like the reference's pinned torch==1.6.0 (sp_baseline.yml:116) instead of
        # following the installed torch's set_to_none=True default -- COCO_Search18 heads then keep decaying / momentum-stepping
        # (default: as conditional_params -- the only models where the two semantics differ are those with conditionally-used
        # parameters, and there the reference's behaviour is the parity target: ADVICE r3)
        self.reference_zero_grad = bool(conditional_params if reference_zero_grad is None else reference_zero_grad)
        self.process_group = process_group
        self.conditional_params = bool(conditional_params)
        with torch.no_grad():
            for p, o in zip(params, offs):
                st = self._dense_strides(p)
                view = torch.as_strided(self.flat_p, p.shape, st, o)
                view.copy_(p.data)
                p.data = view
                p.grad = torch.as_strided(self.flat_g, p.shape, st, o)
                self.state[p] = {"step": torch.zeros((), dtype=torch.float32),
                                 "exp_avg": torch.as_strided(self.flat_m, p.shape, st, o),
                                 "exp_avg_sq": torch.as_strided(self.flat_v, p.shape, st, o)}
        self._bucketer = None
        if process_group is not False:
            from .ddp import GradBucketer, world_size
            # force_bucketer: run the hook -> bucket -> async all-reduce machinery even in a world of one (the single-GPU RCCL test)
            if world_size(process_group) > 1 or (force_bucketer and torch.distributed.is_initialized()):
                torch.distributed.broadcast(self.flat_p, src=0, group=process_group)      # replicas start identical
                if bucket_mb > 0:
                    self._bucketer = GradBucketer(self.flat_g, offs, total, int(bucket_mb * (1 << 20)), process_group)
        for i, p in enumerate(params):
            p.register_post_accumulate_grad_hook(lambda _p, i=i: self._on_grad(i))
            p._sp_flat = _FlatSlot(self, i)      # functional._take_grad_view: backward kernels may write this parameter's gradient in place

    def _on_grad(self, i: int):
        self._touched[i] = True
        self._fresh[i] = False
        self._direct_pending[i] = False
        if self._bucketer is not None and self._params[i].grad.data_ptr() == self.flat_g.data_ptr() + 4 * self._offs[i]:
            self._bucketer.mark_ready(i)      # (a foreign .grad tensor is copied into the flat buffer in step(): no early launch)

    @staticmethod
    def _dense_strides(p):
        # keep the parameter's own (dense, non-overlapping) layout, e.g. channels_last for conv weights
        if p.is_contiguous() or p.dim() != 4:
            return torch.empty(p.shape).stride()
        return torch.empty(p.shape).contiguous(memory_format=torch.channels_last).stride()

    def zero_grad(self, set_to_none: Optional[bool] = None):
        """one memset of the flat buffer; the .grad views always survive (set_to_none only selects WHICH parameters the next
        step() updates, see the class docstring).  Default: the installed torch's set_to_none=True, or the reference's zero-fill
        when the optimizer was built with reference_zero_grad=True."""
        if set_to_none is None:
            set_to_none = not self.reference_zero_grad
        self._sticky = not set_to_none
        if self._bucketer is not None:
            # a backward whose step() was skipped (non-finite-loss guard, caught exception) left buckets launched / counted: wait for
            # the all-reduces in flight before the buffer is cleared, and reset the counters so the next backward is a fresh round
            self._bucketer.drain()
        self.flat_g.zero_()
        self._touched = [False] * len(self._params)
        self._fresh = [True] * len(self._params)
        self._direct_pending = [False] * len(self._params)
        for p, o in zip(self._params, self._offs):
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                p.grad = torch.as_strided(self.flat_g, p.shape, self._dense_strides(p), o)

    def _gather_foreign_grads(self) -> bool:
        """If a caller replaced p.grad (e.g. the stock ``for p in params: p.grad = None`` idiom), copy it back.  Returns whether
        any gradient had to be copied (the bucketed early all-reduce then cannot be used for this step)."""
        foreign = False
        for i, (p, o) in enumerate(zip(self._params, self._offs)):
            g = p.grad
            if g is not None and g.data_ptr() == self.flat_g.data_ptr() + 4 * o:
                continue
            view = torch.as_strided(self.flat_g, p.shape, self._dense_strides(p), o)
            if g is None:
                view.zero_()
                self._touched[i] = False
            else:
                view.copy_(g)
                self._touched[i] = True
                foreign = True
            p.grad = view
        return foreign

    @torch.no_grad()
    def step(self, closure=None):
        for i, pend in enumerate(self._direct_pending):
            if pend:                      # in-place gradient whose hook never fired (see _FlatSlot.done)
                self._on_grad(i)
        foreign = self._gather_foreign_grads()
        g = self.param_groups[0]
        world = 1
        if self.process_group is not False:   # RCCL sum over xGMI of the flat buffer (bucketed, overlapped); averaged in the Adam kernel
            from .ddp import allreduce_sum_, union_flags, world_size
            world = world_size(self.process_group)
            if world > 1 or self._bucketer is not None:
                if self._bucketer is None:
                    allreduce_sum_(self.flat_g, self.process_group)
                elif foreign:
                    raise RuntimeError("FlatAdam: a parameter's .grad was replaced by a foreign tensor while the bucketed, "
                                       "overlapped gradient all-reduce is active; use FlatAdam.zero_grad() (views survive) or "
                                       "construct FlatAdam(bucket_mb=0) for one un-overlapped all-reduce in step()")
                else:
                    self._bucketer.finish()                   # launch what is still pending, wait for every bucket
                if self.conditional_params:   # COCO heads of categories absent from this rank's shard: same decision on every rank
                    self._touched = union_flags(self._touched, self.flat_g.device, self.process_group)
                elif not all(self._touched):
                    missing = sum(1 for t in self._touched if not t)
                    raise RuntimeError(
                        f"FlatAdam: {missing} parameter(s) received no gradient on this rank under data parallelism; construct "
                        "FlatAdam(conditional_params=True) for models with conditionally-used parameters (COCO_Search18 "
                        "per-category heads) so that every rank takes the same decision")
        if self._sticky:
            self._touched = [t or s > 0 for t, s in zip(self._touched, self._steps)]
        L = hip.lib()
        ws = hip.workspace(L.sp_sumsq_workspace(self.numel), self.flat_g.device, slot=1)
        check(L.sp_sumsq(ptr(self.flat_g), self.numel, ptr(self._sumsq), ptr(ws), hip.stream()), "sp_sumsq")
        b1, b2 = g["betas"]
        # torch.optim.Adam skips parameters whose .grad is None (no decay, no moment update, no step increment), e.g.
        # the COCO per-category heads of categories absent from the batch: run the fused kernel per maximal run of
        # consecutive touched parameters that share a step count (one launch for AiR / OSIE).
        n = len(self._params)
        i = 0
        while i < n:
            if not self._touched[i]:
                i += 1
                continue
            j, t = i, self._steps[i] + 1
            while j + 1 < n and self._touched[j + 1] and self._steps[j + 1] + 1 == t:
                j += 1
            lo = self._offs[i]
            hi = self._offs[j + 1] if j + 1 < n else self.numel
            check(L.sp_clip_adam(ptr(self.flat_p) + 4 * lo, ptr(self.flat_g) + 4 * lo, ptr(self.flat_m) + 4 * lo,
                                 ptr(self.flat_v) + 4 * lo, hi - lo, ptr(self._sumsq), 1.0 / world, float(g["clip"]),
                                 float(g["lr"]), b1, b2, float(g["eps"]), float(g["weight_decay"]), 1.0 - b1 ** t,
                                 1.0 - b2 ** t, hip.stream()), "sp_clip_adam")
            for k in range(i, j + 1):
                self._steps[k] = t
                self.state[self._params[k]]["step"] += 1
            i = j + 1
        return self._sumsq.sqrt() / world      # total gradient norm before clipping (device scalar, no sync)

    def load_state_dict(self, state_dict):
        sd = state_dict["state"]
        with torch.no_grad():
            for i, p in enumerate(self._params):
                if i in sd:
                    st = sd[i]
                    self.state[p]["exp_avg"].copy_(st["exp_avg"])
                    self.state[p]["exp_avg_sq"].copy_(st["exp_avg_sq"])
                    self.state[p]["step"].fill_(float(st["step"]))
                    self._steps[i] = int(float(st["step"]))
        for k, v in state_dict["param_groups"][0].items():
            if k != "params":
                self.param_groups[0][k] = v
