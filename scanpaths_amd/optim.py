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


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params: Iterable[torch.nn.Parameter], lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 clip: float = 0.0, process_group=None):
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
        self.process_group = process_group
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
        for i, p in enumerate(params):
            p.register_post_accumulate_grad_hook(lambda _p, i=i: self._touched.__setitem__(i, True))

    @staticmethod
    def _dense_strides(p):
        # keep the parameter's own (dense, non-overlapping) layout, e.g. channels_last for conv weights
        if p.is_contiguous() or p.dim() != 4:
            return torch.empty(p.shape).stride()
        return torch.empty(p.shape).contiguous(memory_format=torch.channels_last).stride()

    def zero_grad(self, set_to_none: bool = False):   # noqa: D401  (views must survive)
        self.flat_g.zero_()
        self._touched = [False] * len(self._params)
        for p, o in zip(self._params, self._offs):
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                p.grad = torch.as_strided(self.flat_g, p.shape, self._dense_strides(p), o)

    def _gather_foreign_grads(self):
        """If a caller replaced p.grad (e.g. the reference's optimizer.zero_grad(set_to_none=True)), copy it back."""
        for i, (p, o) in enumerate(zip(self._params, self._offs)):
            g = p.grad
            view = torch.as_strided(self.flat_g, p.shape, self._dense_strides(p), o)
            if g is None:
                view.zero_()
                self._touched[i] = False
            elif g.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                view.copy_(g)
                self._touched[i] = True
            p.grad = view

    @torch.no_grad()
    def step(self, closure=None):
        self._gather_foreign_grads()
        g = self.param_groups[0]
        world = 1
        if self.process_group is not False:   # RCCL sum over xGMI of ONE flat buffer; averaged inside the Adam kernel
            from .ddp import allreduce_sum_, union_flags
            world = allreduce_sum_(self.flat_g, self.process_group)
            if world > 1:      # e.g. COCO heads of categories absent from this rank's shard; unconditional: a collective
                self._touched = union_flags(self._touched, self.flat_g.device, self.process_group)
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
