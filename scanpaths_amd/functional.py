"""torch.autograd.Function wrappers around the HIP kernels (scanpaths_amd/csrc via hip.py).

Every arithmetic op of the hot path is a kernel launch through the C ABI; torch supplies device memory,
the stream, and the autograd graph (plus pure data movement: cat / stack / views / copies).
Activations are contiguous NHWC tensors ``[N, H, W, C]``; conv weights are nn.Parameters of logical shape
OIHW in ``torch.channels_last`` memory format, i.e. physically ``[Co][KH][KW][Ci]`` -- the layout the
implicit-GEMM kernels consume and the layout their weight gradients are produced in.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
from torch.autograd import Function

from . import config as _config
from . import hip
from .hip import ConvDesc, WgradDesc, check, ptr


# ----------------------------------------------------------------------------------------------------
# raw launches
# ----------------------------------------------------------------------------------------------------
# split-K of the fp32-MFMA pure GEMMs with few output tiles (embeddings, tiny heads): from 8 K-tiles on, 4 K-tiles per split.  The
# kernel has no deep prefetch, so a workgroup's time is its K-tile count times a global-load round trip: the semantic embedding
# (K = 512: 4 workgroups x 16 K-tiles) ran 35 us per launch, 51 launches per step.  A/B in the step (same box, interleaved):
# 298.2 / 298.9 ms against 301.3 / 304.9 with the round-2 rule (from 32 K-tiles on, 8 per split).
_KSPLIT_MIN_NKT, _KSPLIT_KT = 8, 4


def _igemm(X, W, bias, out, *, N_img, Hi, Wi, Kc, ldx, Ho, Wo, Nout, ldc, ldw, KH=1, KW=1, stride=1, pad=0, dil=1,
           mode=0, alpha=1.0, beta=0, relu=0, nbatch=1, sX=0, sW=0, sC=0):
    # a handful of rows against a large weight matrix (the dense layers of the decode loop and their data gradients): the skinny kernel
    # streams the matrix once over ~512 workgroups (csrc/gemm_skinny.hip; 135-185 us per launch on the 128 x 128-tile kernel below)
    M_ = N_img * Ho * Wo
    aligned = all(t is None or t.data_ptr() % 16 == 0 for t in (X, W, out, bias))
    if (SKINNY_GEMM and KH * KW == 1 and nbatch == 1 and M_ <= 64 and not beta and Hi == Ho and Wi == Wo and stride == 1 and pad == 0
            and aligned and ldc % 4 == 0 and hip.lib().sp_gemm_skinny_applies(M_, Nout, Kc, ldx, ldw, ldc, int(mode))):
        L = hip.lib()
        ws = hip.workspace(L.sp_gemm_skinny_workspace(M_, Nout, Kc, int(mode)), X.device, slot=2)

        def launch_sk():
            check(L.sp_gemm_skinny(ptr(X), ptr(W), ptr(bias), ptr(out), M_, Nout, Kc, ldx, ldw, ldc, int(mode), float(alpha), int(relu),
                                   ptr(ws), hip.stream()), "sp_gemm_skinny")
        FUSION_COUNTS["skinny_gemm"] += 1
        if hip.TIMER is None:
            return launch_sk()
        return hip.TIMER.bracket(("skinny_fwd" if mode == 0 else "skinny_dgrad", M_, Nout, Kc, "1x1", 1), 2.0 * M_ * Nout * Kc, launch_sk)
    # split-K when a pure GEMM has too few output tiles to fill 256 CUs (e.g. M = batch rows, K = 13824)
    ksplit, ws = 0, None
    if KH * KW == 1 and nbatch == 1:
        tiles = ((N_img * Ho * Wo + 127) // 128) * ((Nout + 127) // 128)
        nkt = (Kc + 31) // 32
        if tiles <= 64 and nkt >= _KSPLIT_MIN_NKT:
            ksplit = max(2, min(nkt // _KSPLIT_KT, 512 // tiles))
            ws = hip.workspace(ksplit * N_img * Ho * Wo * Nout * 4, X.device, slot=2)
    d = ConvDesc(N_img, Hi, Wi, Kc, ldx, Ho, Wo, Nout, ldc, KH, KW, stride, pad, dil, mode, ldw, float(alpha), int(beta),
                 int(relu), nbatch, sX, sW, sC, ksplit, ptr(ws))

    def launch():
        check(hip.lib().sp_conv_igemm(C.byref(d), ptr(X), ptr(W), ptr(bias), ptr(out), hip.stream()), "sp_conv_igemm")
    if hip.TIMER is None:
        return launch()
    M = N_img * Ho * Wo
    K = KH * KW * Kc
    key = ("igemm_fwd" if mode == 0 else "igemm_dgrad", M, Nout, K, f"{KH}x{KW}", nbatch)
    hip.TIMER.bracket(key, 2.0 * M * Nout * K * nbatch, launch)


def _wgrad(X, dY, dW, *, N_img, Hi, Wi, Ci, ldx, Ho, Wo, Co, ldy, ldo, KH=1, KW=1, stride=1, pad=0, dil=1, beta=0,
           alpha=1.0, nbatch=1, sX=0, sY=0, sO=0):
    d = WgradDesc(N_img, Hi, Wi, Ci, ldx, Ho, Wo, Co, ldy, KH, KW, stride, pad, dil, ldo, int(beta), float(alpha), nbatch,
                  sX, sY, sO)
    L = hip.lib()
    ws = hip.workspace(L.sp_conv_wgrad_workspace(C.byref(d)), X.device, slot=0)
    def launch():
        check(L.sp_conv_wgrad(C.byref(d), ptr(X), ptr(dY), ptr(dW), ptr(ws), hip.stream()), "sp_conv_wgrad")
    if hip.TIMER is None:
        return launch()
    M = N_img * Ho * Wo
    key = ("wgrad", M, Co, KH * KW * Ci, f"{KH}x{KW}", nbatch)
    hip.TIMER.bracket(key, 2.0 * M * Co * KH * KW * Ci * nbatch, launch)


# ---- fp32-faithful GEMMs on the 16-bit matrix pipe (csrc/conv_f16x2.hip, csrc/conv_bf16x3.hip) ---------------------
# USE_BF16X3 False -> every GEMM on the fp32 MFMA kernel (parity triage / A-B timing).
# SPLIT_SCHEME "f16x2": 2 fp16 planes + power-of-two scales, 3 MFMA products (default);
#              "bf16x3": 3 bf16 planes, 6 products (no scale pass; also used when a channel count is not a multiple of 32).
# THROUGHPUT_MODE (config split_scheme="f16x1", bench.py --precision f16x1, never the default): the f16x2 operand storage and kernels
# with only the main product, i.e. GEMM operands rounded to one fp16 plane, fp32 accumulation.  ~2^-11 relative GEMM error: fails the
# parity bar.  All of these and the fusion switches below are module globals set from scanpaths_amd.config (ONE switchboard, set in
# code; environment variables only under SP_ALLOW_ENV_TUNING=1) -- tests monkeypatch the globals directly.
def _apply_config():
    g = globals()
    c = _config.settings
    g["FUSED_AMAX"] = bool(c["fused_amax"])      # producers leave max|output| behind for the operand split (see _amax_hint)
    g["USE_BF16X3"] = bool(c["use_split"])
    g["THROUGHPUT_MODE"] = c["split_scheme"] == "f16x1"
    g["SPLIT_SCHEME"] = "f16x2" if c["split_scheme"] == "f16x1" else c["split_scheme"]
    for name, key in (("GRAD_MERGE", "grad_merge"), ("BN_SPLIT", "bn_split"), ("BN_SKIP_DX", "bn_skip_dx"), ("BN_SKIP_Z", "bn_skip_z"),
                      ("LSTM_BWD_SPLIT", "lstm_bwd_split"), ("RANK1_DSP_SPLIT", "rank1_dsp_split"), ("RANK1_DWC_SPLIT", "rank1_dwc_split"),
                      ("LSTM_SKIP_DPRE", "lstm_skip_dpre"), ("FUSE_GATE_LSTM", "fuse_gate_lstm"), ("LSTM_H_PLANES", "lstm_h_planes"),
                      ("DEFER_WGRAD", "defer_wgrad"), ("CHANNEL_SCALES", "channel_scales"), ("HW2_SINGLE", "hw2_single"),
                      ("ROW_SPARSITY", "row_sparsity"), ("DIRECT_GRAD", "direct_grad"), ("SKINNY_GEMM", "skinny_gemm"), ("ASYNC_DGRAD", "async_dgrad"),
                      ("DRT_BATCHED", "drt_batched"), ("RANK1_FUSED", "rank1_fused")):
        g[name] = bool(c[key])


_apply_config()


# how often each fused pass ran (tests assert that the benchmark's kernel path, not a fallback, is the one under test)
FUSION_COUNTS = {"lstm_bwd_split": 0, "bn_bwd_split_operand": 0, "gateconv_lstm": 0, "gateconv_lstm_hplanes": 0,
                 "bn_fwd_split": 0, "bn_fwd_split_operand": 0, "bn_skip_z": 0, "bn_bwd_split": 0, "bn_skip_dx": 0,
                 "conv_bn_stats": 0, "grad_merge": 0, "rank1_dsp_split": 0, "rank1_dwc_split": 0, "lstm_skip_dpre": 0, "wgrad_multi": 0,
                 "row_sparse_bwd": 0, "fan_in_rows": 0, "direct_grad": 0, "output_gate": 0, "skinny_gemm": 0, "async_dgrad": 0,
                 "drt_fwd_batched": 0, "rank1_fused": 0}


# ---- parameter gradients written straight into the optimizer's flat gradient buffer -------------------------------------------------
# FlatAdam gives every parameter a view of ONE flat gradient buffer as its .grad and marks it (`_sp_flat`).  A backward kernel that
# produces the gradient of such a LEAF parameter, and is the first to do so since FlatAdam.zero_grad() cleared the buffer, writes into
# that view and hands autograd None: the ~160 per-parameter read-modify-write adds of AccumulateGrad (and as many temporaries) per
# training step go away.  The parameter's AccumulateGrad node still runs (with an undefined gradient) and fires the optimizer's
# post-accumulate hook (gradient-ready bookkeeping, bucketed all-reduce) once all contributions of the backward pass exist.
# A second gradient for the same parameter before the next zero_grad() takes the ordinary autograd path and is added.
def _grad_slot(param):
    """(parameter, its AccumulateGrad node) of a leaf parameter whose .grad is a FlatAdam view, else None"""
    if not DIRECT_GRAD or param is None or not getattr(param, "is_leaf", False) or not param.requires_grad:
        return None
    if getattr(param, "_sp_flat", None) is None:
        return None
    if getattr(param, "_backward_hooks", None):      # a user tensor hook (p.register_hook: scaling, logging) must see the gradient: autograd's path
        return None
    try:
        return param, torch.autograd.graph.get_gradient_edge(param).node
    except Exception:          # (a torch without gradient edges: ordinary path)
        return None


def _take_grad_view(pn, phys_perm=None):
    """-> (slot, tensor to write the gradient into) when the gradient may be written in place, else (None, None).
    pn: what _grad_slot returned in forward.  phys_perm: permutation that must make the view contiguous (conv weights: (0, 2, 3, 1)).
    Only inside a backward pass that ACCUMULATES into this parameter (loss.backward()): under torch.autograd.grad(), or
    backward(inputs=...) without it, the engine does not run the parameter's AccumulateGrad node and the caller wants the gradient
    returned -- writing it into .grad there would be a silent side effect."""
    if pn is None:
        return None, None
    param, node = pn
    try:
        if not torch._C._will_engine_execute_node(node):
            return None, None
    except RuntimeError:       # "a leaf node was passed ... running autograd.grad()"
        return None, None
    slot = param._sp_flat
    g = slot.peek(param)
    if g is None:
        return None, None
    v = g.permute(*phys_perm) if phys_perm is not None else g
    if not v.is_contiguous():
        return None, None
    slot.take()
    FUSION_COUNTS["direct_grad"] += 1
    return slot, v


# ---- masked-step sparsity of the backward pass, derived from the gradient that reaches the model's outputs ----------------------------
# The reference multiplies every loss term by action_masks / duration_masks (AiR/models/loss.py:10-14,27-32; AiR/train.py:190-197): a
# sample whose scanpath ended at step L sends no gradient into any prediction of a decode step t > L, nothing of sample b reaches
# another sample inside the decoder (no BatchNorm there, per-sample attention memories), so the whole backward recurrence of
# (b, t > L_b) is EXACTLY zero.  The reference computes those zeros densely.  Here nobody has to promise anything: decode() hands its
# outputs through ONE identity node (_OutputGate).  Autograd runs that node's backward once the gradients of ALL outputs exist -- from
# whatever consumed them: the fused loss, the reference's two separate loss calls, an auxiliary loss, the RL losses -- and before any
# node of the decoder.  It computes last[b] = the last decode step at which any output gradient of sample b is non-zero (sp_rows_last:
# device, no host sync; NaN counts as non-zero, a gradient that never arrived as zero) and leaves it on the forward's own token
# (DecodeRows); the step-tagged backward kernels of THAT decode read it from the token they were built with (DecodeStep).  A consumer
# that reads every step makes last[b] = T - 1 and the pass dense "because the gradient says so"; two forwards inside one backward
# carry two tokens; a retained graph recomputes the context at the gate of every backward.  (One semantic difference from multiplying
# the zeros: a sample whose incoming gradient is exactly zero behind step L but whose ACTIVATIONS there are inf / NaN gets zero instead
# of NaN gradient contributions from those steps.)
class RowsCtx:
    __slots__ = ("last", "B")

    def __init__(self, last):
        self.last, self.B = last, int(last.numel())


class DecodeRows:
    """token of one decode() call: .rc = the RowsCtx of the backward pass in flight through that decode's graph (set by _OutputGate)"""
    __slots__ = ("rc",)

    def __init__(self):
        self.rc = None

    def at(self, t: int) -> "DecodeStep":
        return DecodeStep(t, self)


class DecodeStep(int):
    """decode step (or memory-update) index t that also knows which decode() it belongs to: what the step-tagged ops take as `step=`"""
    def __new__(cls, t, rows):
        o = int.__new__(cls, t)
        o.rows = rows
        return o


def rows_ctx(step, nsamples):
    """the row-sparsity context of the backward pass in flight, if this op was tagged with a DecodeStep of a batch of nsamples samples"""
    tok = getattr(step, "rows", None)
    if tok is None or not ROW_SPARSITY:
        return None
    rc = tok.rc
    if rc is None or rc.B != nsamples:
        return None
    return rc


class _OutputGate(Function):
    """identity over the stacked outputs of decode() ([nstack, B, T, ...] each); its backward derives last[b] from the incoming gradients"""
    @staticmethod
    def forward(ctx, tok, *outs):
        ctx.tok = tok
        ctx.set_materialize_grads(False)          # an output nobody consumed contributes None, not a zero tensor to scan
        return tuple(o.view_as(o) for o in outs)

    @staticmethod
    def backward(ctx, *grads):
        tok = ctx.tok
        tok.rc = None
        live = [g for g in grads if g is not None]
        if ROW_SPARSITY and live:
            gs = [None if g is None else g.contiguous() for g in grads]
            nstack, B, T = live[0].shape[:3]
            for g in gs:
                if g is not None and (g.dtype != torch.float32 or tuple(g.shape[:3]) != (nstack, B, T)):
                    return (None,) + tuple(grads)              # (nothing this build produces; stay dense rather than guess)
            n = len(gs)
            arr = (C.c_void_p * n)(*[None if g is None else ptr(g) for g in gs])
            rl = (C.c_int64 * n)(*[0 if g is None else g.numel() // (nstack * B * T) for g in gs])
            last = torch.empty(B, dtype=torch.int32, device=live[0].device)
            check(hip.lib().sp_rows_last(arr, rl, n, nstack, B, T, ptr(last), hip.stream()), "sp_rows_last")
            tok.rc = RowsCtx(last)
            FUSION_COUNTS["output_gate"] += 1
            grads = gs
        return (None,) + tuple(grads)


def output_gate(tok: DecodeRows, outs):
    """outs: list of decode() output stacks -> the same tensors behind the gate node (see the block comment above)"""
    return list(_OutputGate.apply(tok, *outs))


def reset_fusion_counts():
    for k in FUSION_COUNTS:
        FUSION_COUNTS[k] = 0


class SplitOperand:
    """a GEMM operand in split form: buf (16-bit planes, interleaved per 16 k) + device scale (f16x2 only).
    kind (f16x2): "scalar" -- one power-of-two scale for the tensor (scale = {scale, amax} words);
                  "cols"   -- activations / gradients [rows][C] with one scale per CHANNEL (scale = [C] vector): exact for the weight
                              gradient (K = pixels); in a forward / data-gradient GEMM the weight operand must absorb the vector;
                  "rows"   -- weights with one scale per row = output column of the GEMM (scale = [rows] vector); `absorbed` names the
                              channel-scale vector the rows were divided by (None: none)."""
    __slots__ = ("buf", "scale", "scheme", "kind", "absorbed", "rows")

    def __init__(self, buf, scale, scheme, kind="scalar", absorbed=None):
        self.buf, self.scale, self.scheme, self.kind, self.absorbed = buf, scale, scheme, kind, absorbed
        self.rows = None      # (RowsCtx, decode step) of a gradient operand whose samples behind their last loss step are exactly zero


def _scheme_for(kc: int) -> str:
    return "f16x2" if SPLIT_SCHEME == "f16x2" and kc % 32 == 0 else "bf16x3"


def split3(x: torch.Tensor) -> torch.Tensor:
    """fp32 [..., K] (K % 16 == 0) -> "split-3 interleaved" bf16 operand [..., K/16, 3, 16] with x = p0+p1+p2 to ~2^-27."""
    x = x.contiguous()
    n = x.numel()
    out = torch.empty(3 * n + 32, dtype=torch.bfloat16, device=x.device)      # + 64-byte zero block for masked loader lanes
    check(hip.lib().sp_split3_bf16(ptr(x), n, ptr(out), hip.stream()), "sp_split3_bf16")
    return out


def split3_wT(wp: torch.Tensor) -> torch.Tensor:
    """physical weight [Co,KH,KW,Ci] -> split-3 operand with rows ci and k = (tap, co)  (dgrad B operand)."""
    Co, KH, KW, Ci = wp.shape
    out = torch.empty(3 * wp.numel() + 32, dtype=torch.bfloat16, device=wp.device)
    check(hip.lib().sp_split3_bf16_wT(ptr(wp), Co, KH * KW, Ci, ptr(out), hip.stream()), "sp_split3_bf16_wT")
    return out


_HINT_POOL = {}


def _amax_hint(device) -> Optional[torch.Tensor]:
    """{scale, amax bits} buffer for a producer kernel that leaves max|output| behind (block max + one guarded atomic per block);
    attached to the produced tensor as ``_sp_amax`` so that split_op can skip its read-only amax pass.  None when the 2xfp16
    back-end is not in use."""
    if not USE_BF16X3 or SPLIT_SCHEME != "f16x2" or not FUSED_AMAX:
        return None
    key = device.index if device.index is not None else torch.cuda.current_device()
    pool = _HINT_POOL.get(key)
    # A pool created while a stream is being captured is zero-filled by a graph node: EVERY replay re-zeroes all of it.  Such a pool
    # may therefore only serve captures (whose launchers reset each slot in stream order, sp_set_tuning("amax_reset")); eager
    # code that drew its single-use slots from it would see them wiped -- or filled -- by a later replay.
    capturing = torch.cuda.is_current_stream_capturing()
    if pool is None or pool[1] >= pool[0].shape[0] or (pool[2] and not capturing):
        pool = [torch.zeros((2048, 2), dtype=torch.float32, device=device), 0, capturing]      # one zero-fill per 2048 hints
        _HINT_POOL[key] = pool
    # an independent tensor over the pool's storage, NOT a view: views share one version counter, so a single in-place torch op on any
    # slot would make autograd reject every slot saved for backward (operand scales are)
    hint = torch.empty(0, dtype=torch.float32, device=device).set_(pool[0].untyped_storage(), 2 * pool[1], (2,))
    pool[1] += 1
    return hint


def _reserve_hints(device, n: int) -> None:
    """make sure the next n slots come from a pool that exists already (created, i.e. zero-filled, under the CURRENT stream): called
    before work is enqueued on another stream that may draw slots, so that no pool is ever created there"""
    if not _amax_hint_active():
        return
    key = device.index if device.index is not None else torch.cuda.current_device()
    pool = _HINT_POOL.get(key)
    capturing = torch.cuda.is_current_stream_capturing()
    if pool is None or pool[1] + n > pool[0].shape[0] or (pool[2] and not capturing):
        _HINT_POOL[key] = [torch.zeros((2048, 2), dtype=torch.float32, device=device), 0, capturing]


_ONES = {}


def _one(device) -> torch.Tensor:
    """device scalar 1.0f: the x_scale of an operand whose (per-channel) scales were absorbed by the weight operand"""
    key = device.index if device.index is not None else torch.cuda.current_device()
    t = _ONES.get(key)
    if t is None:
        t = _ONES[key] = torch.ones(2, dtype=torch.float32, device=device)
    return t


def _amax_hint_active() -> bool:
    return bool(USE_BF16X3 and SPLIT_SCHEME == "f16x2" and FUSED_AMAX)


def _scale_slot(device) -> torch.Tensor:
    """{scale, amax scratch} words for a split pass that computes max|x| itself: a ZEROED single-use slot of the hint pool (the amax
    kernel's atomicMax needs a zero word; with sp_set_tuning("amax_reset", 1) the launcher then adds its one-thread reset kernel only
    under stream capture -- ~160 launches per training step otherwise)"""
    slot = _amax_hint(device)
    return slot if slot is not None else torch.zeros(2, dtype=torch.float32, device=device)


def _hint_ptr(hint: Optional[torch.Tensor]) -> Optional[int]:
    return None if hint is None else hint.data_ptr() + 4


def _fp32_required(x: torch.Tensor, what: str) -> None:
    """a tensor whose fp32 form was left unwritten by its producer (bn skip_dx, cell backward skip) must never be read as fp32"""
    if getattr(x, "_sp_skipped", False):
        raise RuntimeError(f"scanpaths_amd: {what} needs the fp32 form of a gradient that exists only as a split operand; "
                           "run with SP_LSTM_SKIP_DPRE=0")




def split_op(x: torch.Tensor, scheme: Optional[str] = None, channel: bool = False) -> SplitOperand:
    """fp32 [..., K] -> split operand; the scheme follows the row length unless given (both operands of a GEMM must agree).
    channel (2xfp16 only): the operand is an activation / gradient [rows][C] of a conv -- when no producer left a max|.| hint (the
    split needs its own pass over x anyway) it gets one power-of-two scale per CHANNEL instead of one per tensor (kind "cols"): a
    channel far below the tensor's maximum (the terminate-logit columns of the saliency tap gradient: 2^-20 .. 2^-29 of the action
    map's) keeps its 22 bits in the weight-gradient GEMM, whose output row / column it alone feeds."""
    scheme = scheme or _scheme_for(x.shape[-1])
    cache = getattr(x, "_sp_cache", None)        # {scheme: SplitOperand} shared by every alias of a multi-consumer tensor (fanout)
    if cache is not None and scheme in cache:
        return cache[scheme]
    _fp32_required(x, "split_op")
    if scheme == "bf16x3":
        op = SplitOperand(split3(x), None, scheme)
    else:
        hint = getattr(x, "_sp_amax", None)      # left by the kernel that produced x (same tensor object, never modified since)
        xc = x.contiguous()
        n = xc.numel()
        out = torch.empty(2 * n + 32, dtype=torch.float16, device=x.device)
        Cc = xc.shape[-1]
        if channel and CHANNEL_SCALES and hint is None and xc.dim() >= 2 and Cc % 16 == 0 and not THROUGHPUT_MODE:
            cscale = torch.empty(Cc, dtype=torch.float32, device=x.device)
            L = hip.lib()
            ws = hip.workspace(L.sp_split2_f16_cols_workspace(n // Cc, Cc), x.device, slot=1)
            check(L.sp_split2_f16_cols(ptr(xc), n // Cc, Cc, ptr(out), ptr(cscale), ptr(ws), hip.stream()), "sp_split2_f16_cols")
            op = SplitOperand(out, cscale, scheme, "cols")
        else:
            scale = hint if hint is not None else _scale_slot(x.device)
            check(hip.lib().sp_split2_f16(ptr(xc), n, ptr(out), ptr(scale), int(hint is not None), hip.stream()), "sp_split2_f16")
            op = SplitOperand(out, scale, scheme)
    if cache is not None:                        # set by producers whose output feeds several GEMMs (the ConvLSTM state h)
        cache[scheme] = op
    return op


def _absorb_of(act: Optional[SplitOperand]) -> Optional[torch.Tensor]:
    """the channel-scale vector a weight operand must absorb to meet `act` in a GEMM that contracts over channels"""
    return act.scale if (act is not None and act.kind == "cols") else None


def split_w(w2d: torch.Tensor, scheme: str, Kc: Optional[int] = None, absorb: Optional[torch.Tensor] = None) -> SplitOperand:
    """weights [rows][K] (K = taps * Kc contiguous) as the [N][K] operand of a forward GEMM: one scale per row (2xfp16), the rows
    divided by the activation's per-channel scales `absorb` [Kc] when it has them"""
    if scheme == "bf16x3":
        return SplitOperand(split3(w2d), None, scheme)
    w2d = w2d.contiguous()
    K = w2d.shape[-1]
    rows = w2d.numel() // K
    out = torch.empty(2 * w2d.numel() + 32, dtype=torch.float16, device=w2d.device)
    rscale = torch.empty(rows, dtype=torch.float32, device=w2d.device)
    check(hip.lib().sp_split2_f16_rows(ptr(w2d), rows, K, Kc if Kc is not None else K, ptr(absorb), ptr(out), ptr(rscale), hip.stream()),
          "sp_split2_f16_rows")
    return SplitOperand(out, rscale, scheme, "rows", absorb)


def split_op_wT(wp: torch.Tensor, scheme: Optional[str] = None, absorb: Optional[torch.Tensor] = None) -> SplitOperand:
    """physical weight [Co,KH,KW,Ci] -> the data gradient's operand: rows ci, k = (tap, co), one scale per row (2xfp16), the values
    divided by the gradient operand's per-channel scales `absorb` [Co] when it has them"""
    Co, KH, KW, Ci = wp.shape
    scheme = scheme or _scheme_for(Co)
    if scheme == "bf16x3":
        return SplitOperand(split3_wT(wp), None, scheme)
    out = torch.empty(2 * wp.numel() + 32, dtype=torch.float16, device=wp.device)
    if Ci % 4 == 0:
        rscale = torch.empty(Ci, dtype=torch.float32, device=wp.device)
        check(hip.lib().sp_split2_f16_wT_rows(ptr(wp), Co, KH * KW, Ci, ptr(absorb), ptr(out), ptr(rscale), hip.stream()),
              "sp_split2_f16_wT_rows")
        return SplitOperand(out, rscale, scheme, "rows", absorb)
    assert absorb is None
    scale = _scale_slot(wp.device)
    check(hip.lib().sp_split2_f16_wT(ptr(wp), Co, KH * KW, Ci, ptr(out), ptr(scale), hip.stream()), "sp_split2_f16_wT")
    return SplitOperand(out, scale, scheme)


def _weight_operand(wp, act: SplitOperand, wcache, transposed=False) -> SplitOperand:
    """the split form of the physical weight wp [Co,KH,KW,Ci] that meets the activation-side operand `act` (forward: [Co][K] rows;
    transposed: the data gradient's [Ci][K]); cached in wcache only when it absorbed no per-channel scales (those belong to ONE tensor)"""
    absorb = _absorb_of(act)
    key = ("wT" if transposed else "w", act.scheme)
    if absorb is None and wcache is not None and key in wcache:
        return wcache[key]
    if transposed:
        op = split_op_wT(wp, act.scheme, absorb)
    else:
        Co, KH, KW, Ci = wp.shape
        op = split_w(wp.reshape(Co, KH * KW * Ci), act.scheme, Ci, absorb)
    if absorb is None and wcache is not None:
        wcache[key] = op
    return op


# Parity tests run the benchmark's kernel path at a small batch: the cost models below then price every GEMM as if its pixel
# dimension were COST_M_SCALE times larger (bs 2 with scale 16 takes exactly the decisions of bs 32).  1.0 in the product path.
COST_M_SCALE = 1.0


def _b3_pays(M, N, K, Kc, nbatch=1, a_elems=None, free_a=False):
    """cost model: split passes (10-12 B per operand element at ~4 TB/s) + split GEMM (~2.3x / ~4x the fp32 rate) < fp32 GEMM.
    a_elems: elements of the activation-side tensor (the loaders use 32-bit byte offsets into the split operand)"""
    if not USE_BF16X3 or nbatch != 1 or Kc % 16 or N < 64:
        return False
    f16 = _scheme_for(Kc) == "f16x2"
    Mc = M * COST_M_SCALE
    flops = 2.0 * Mc * N * K
    split_bytes = (12.0 if f16 else 10.0) * ((0 if free_a else Mc * Kc) + N * K)     # free_a: the activation split already exists
    bpe = 4.0 if f16 else 6.0
    if bpe * (a_elems if a_elems is not None else 4.0 * M * Kc) + 64 >= 2 ** 32 or bpe * N * K + 64 >= 2 ** 32:
        return False
    return flops * (1 / 1.1e14 - 1 / (4.0e14 if f16 else 2.5e14)) > split_bytes / 4e12 and flops > (1e9 if free_a else 2e9)


def _igemm_b3(Xs, Ws, bias, out, *, N_img, Hi, Wi, Kc, ldx, Ho, Wo, Nout, ldc, ldw, KH=1, KW=1, stride=1, pad=0, dil=1, mode=0,
              alpha=1.0, beta=0, relu=0, stats=None, rows=None):
    """implicit-GEMM conv / dgrad on split operands (SplitOperand, or a raw split-3 buffer)"""
    if not isinstance(Xs, SplitOperand):
        Xs, Ws = SplitOperand(Xs, None, "bf16x3"), SplitOperand(Ws, None, "bf16x3")
    assert Xs.scheme == Ws.scheme, (Xs.scheme, Ws.scheme)
    d = ConvDesc(N_img, Hi, Wi, Kc, ldx, Ho, Wo, Nout, ldc, KH, KW, stride, pad, dil, mode, ldw, float(alpha), int(beta),
                 int(relu), 1, 0, 0, 0, 0, None)
    f16 = Xs.scheme == "f16x2"
    xscale = Xs.scale
    if f16 and rows is not None and mode == 1:      # (RowsCtx, step): zero tiles for samples without loss gradient at this decode step
        d.row_last, d.row_step = rows[0].last.data_ptr(), int(rows[1])
    if f16:
        d.w_scale_rows = int(Ws.kind == "rows")
        if Xs.kind == "cols":      # the per-channel scales of the activation operand live in the weight operand
            if Ws.absorbed is not Xs.scale:
                raise RuntimeError("scanpaths_amd: a per-channel-scaled operand met a weight operand that did not absorb its scales")
            xscale = _one(out.device)
        elif Ws.absorbed is not None:
            raise RuntimeError("scanpaths_amd: a weight operand that absorbed per-channel scales met another activation operand")

    def launch():
        if stats is not None:          # (partial, mm): BatchNorm statistics of the output from the epilogue
            check(hip.lib().sp_conv_igemm_f16x2_stats(C.byref(d), ptr(Xs.buf), ptr(xscale), ptr(Ws.buf), ptr(Ws.scale), ptr(out),
                                                      ptr(stats[0]), ptr(stats[1]), hip.stream()), "sp_conv_igemm_f16x2_stats")
        elif f16:
            fn = hip.lib().sp_conv_igemm_f16x1 if THROUGHPUT_MODE else hip.lib().sp_conv_igemm_f16x2
            check(fn(C.byref(d), ptr(Xs.buf), ptr(xscale), ptr(Ws.buf), ptr(Ws.scale), ptr(bias), ptr(out), hip.stream()),
                  "sp_conv_igemm_f16x2")
        else:
            check(hip.lib().sp_conv_igemm_bf16x3(C.byref(d), ptr(Xs.buf), ptr(Ws.buf), ptr(bias), ptr(out), hip.stream()),
                  "sp_conv_igemm_bf16x3")
    if hip.TIMER is None:
        return launch()
    M = N_img * Ho * Wo
    K = KH * KW * Kc
    pre = ("h1" if THROUGHPUT_MODE else "h2") if f16 else "b3"
    sparse = f16 and rows is not None and mode == 1      # "_rows": the launch skips samples without loss gradient (bench.py scales its FLOPs)
    key = (pre + ("_fwd" if mode == 0 else "_dgrad") + ("_rows" if sparse else ""), M, Nout, K, f"{KH}x{KW}", 1)
    hip.TIMER.bracket(key, 2.0 * M * Nout * K, launch)


def _wgrad_scheme(Ci, Co) -> str:
    return "f16x2" if _scheme_for(Ci) == "f16x2" and _scheme_for(Co) == "f16x2" else "bf16x3"


def _w3_pays(M, Co, K, Ci, nbatch=1, free_splits=False):
    """free_splits: both split operands already exist (x kept from the forward GEMM, dy split for the data-gradient GEMM), so
    the split weight-gradient kernel only has to beat the fp32 one"""
    f16 = _wgrad_scheme(Ci, Co) == "f16x2"          # the 2xfp16 kernel takes any Ci % 16 == 0, the 3xbf16 one needs Ci % 128 == 0
    if not USE_BF16X3 or nbatch != 1 or Co % 16 or Co < 64 or (Ci % 128 and not f16):
        return False
    flops = 2.0 * M * COST_M_SCALE * Co * K
    split_bytes = 0.0 if free_splits else (12.0 if f16 else 10.0) * M * COST_M_SCALE * (Ci + Co)
    if 6.0 * M * max(Ci, Co) * 4 >= 2 ** 32:
        return False
    return flops * (1 / 1.0e14 - 1 / (3.2e14 if f16 else 1.6e14)) > split_bytes / 4e12 and flops > 2e9


HW2_SINGLE_MIN_FLOPS = 9e10      # smallest single weight gradient routed to hw2_kernel (256 x 256 tiles, single-level accumulation over <= 20480 pixels): the x-gate conv, sal_conv and -- round 4, tools/encoder_census.py --hw2-min-flops: -2.0 ms per step -- the encoder's layer-3 / layer-4 weight gradients from Co x K = 256 x 2304 on (below that the 256 KB slab tiles cost more than the larger tile saves)


def _wgrad_b3(Xs, dYs, dW, *, N_img, Hi, Wi, Ci, Ho, Wo, Co, ldo, KH=1, KW=1, stride=1, pad=0, dil=1, beta=0, alpha=1.0,
              ws_slot=0):
    if not isinstance(Xs, SplitOperand):
        Xs, dYs = SplitOperand(Xs, None, "bf16x3"), SplitOperand(dYs, None, "bf16x3")
    assert Xs.scheme == dYs.scheme, (Xs.scheme, dYs.scheme)
    d = WgradDesc(N_img, Hi, Wi, Ci, Ci, Ho, Wo, Co, Co, KH, KW, stride, pad, dil, ldo, int(beta), float(alpha), 1, 0, 0, 0)
    L = hip.lib()
    f16 = Xs.scheme == "f16x2"
    if f16:      # K = pixels: per-channel scales of either operand factor out in the epilogue / slab reduce
        d.x_scale_vec, d.y_scale_vec = int(Xs.kind == "cols"), int(dYs.kind == "cols")
    M = N_img * Ho * Wo
    # large single weight gradients whose shape fits the 256 x 256-tile kernel (x-gate conv, sal_conv): hw2_kernel with one segment
    big = L.sp_conv_wgrad_f16x2_multi_workspace(C.byref(d), 1) if (f16 and HW2_SINGLE and not THROUGHPUT_MODE
                                                                   and 2.0 * M * COST_M_SCALE * Co * KH * KW * Ci >= HW2_SINGLE_MIN_FLOPS) else 0
    ws = hip.workspace(big if big > 0 else (L.sp_conv_wgrad_f16x2_workspace if f16 else L.sp_conv_wgrad_bf16x3_workspace)(C.byref(d)),
                       dW.device, slot=ws_slot)

    def launch():
        if big > 0:
            FUSION_COUNTS["wgrad_multi"] += 1
            one = lambda t: (C.c_void_p * 1)(t.data_ptr())
            check(L.sp_conv_wgrad_f16x2_multi(C.byref(d), 1, one(Xs.buf), one(Xs.scale), one(dYs.buf), one(dYs.scale), ptr(dW), ptr(ws),
                                              None, None, hip.stream()), "sp_conv_wgrad_f16x2_multi")
        elif f16:
            fn = L.sp_conv_wgrad_f16x1 if THROUGHPUT_MODE else L.sp_conv_wgrad_f16x2
            check(fn(C.byref(d), ptr(Xs.buf), ptr(Xs.scale), ptr(dYs.buf), ptr(dYs.scale), ptr(dW), ptr(ws), hip.stream()),
                  "sp_conv_wgrad_f16x2")
        else:
            check(L.sp_conv_wgrad_bf16x3(C.byref(d), ptr(Xs.buf), ptr(dYs.buf), ptr(dW), ptr(ws), hip.stream()),
                  "sp_conv_wgrad_bf16x3")
    if hip.TIMER is None:
        return launch()
    hip.TIMER.bracket(((("h1" if THROUGHPUT_MODE else "h2") if f16 else "b3") + ("_wgrad_multi" if big > 0 else "_wgrad"), M, Co, KH * KW * Ci, f"{KH}x{KW}", 1), 2.0 * M * Co * KH * KW * Ci, launch)


def colsum(x2d: torch.Tensor, C_: int, ld: int, M: int) -> torch.Tensor:
    out = torch.empty(C_, dtype=torch.float32, device=x2d.device)
    L = hip.lib()
    ws = hip.workspace(L.sp_colsum_workspace(M, C_), x2d.device, slot=1)
    check(L.sp_colsum(ptr(x2d), M, C_, ld, ptr(out), 0, ptr(ws), hip.stream()), "sp_colsum")
    return out


def _colsum_any(x: torch.Tensor, C_: int) -> torch.Tensor:
    """sum over all leading dims of a contiguous [..., C_] tensor; pads C_ to a multiple of 4 when needed."""
    M = x.numel() // C_
    if C_ % 4 == 0:
        return colsum(x, C_, C_, M)
    Cp = (C_ + 3) // 4 * 4
    xp = torch.empty((M, Cp), dtype=torch.float32, device=x.device)
    check(hip.lib().sp_pad_lastdim(ptr(x), M, C_, Cp, ptr(xp), hip.stream()), "sp_pad_lastdim")
    return colsum(xp, Cp, Cp, M)[:C_]


def _add_raw(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty_like(a)
    check(hip.lib().sp_add(ptr(a), ptr(b), ptr(out), a.numel(), hip.stream()), "sp_add")
    return out


class _Add(Function):
    @staticmethod
    def forward(ctx, a, b):
        return _add_raw(a, b)

    @staticmethod
    def backward(ctx, g):
        return g, g


class _ZeroGradTouch(Function):
    """Identity on ``t`` that makes ``params`` part of the graph with an exactly-zero gradient.
    Used for parameters whose contribution cancels analytically (the attention "cur" branches and biases,
    baseline_attention.py:82-86,117-121): the reference gives them a ~1e-17 gradient, hence weight decay and an
    Adam step; a zero gradient (instead of None) reproduces that with any optimizer."""
    @staticmethod
    def forward(ctx, t, *params):
        ctx.shapes = [(p.shape, p.stride(), p.device) for p in params]
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        zeros = tuple(torch.zeros(sh, dtype=torch.float32, device=dev) for sh, _, dev in ctx.shapes)
        return (g,) + zeros


def touch_zero_grad(t, params):
    return _ZeroGradTouch.apply(t, *params)


class _FanOut(Function):
    """x -> n aliases of x for n consumers; backward adds the n gradients in ONE pass (sp_sum_n) instead of the n-1
    read-read-write adds autograd would issue.  Used for the hoisted x-gate pre-activations (one consumer per decode step).
    token: dict shared with the aliases (alias i carries ``_sp_fan = (token, i)``).  A consumer whose backward leaves the fp32 form of
    its gradient UNWRITTEN (the cell backward, _lstm_rank1_backward) records the split operand it wrote instead under its alias index;
    this backward then reads that contribution from the token BY OUTPUT INDEX -- never from an attribute of the incoming tensor, whose
    identity autograd does not guarantee (a hook that returns a new tensor, an accumulation, a cloning wrapper: ADVICE r3) -- and
    raises if a recorded contribution's gradient never arrived or arrives marked as unwritten without a record."""
    @staticmethod
    def forward(ctx, x, n, token, step=None, events=None):
        ctx.n, ctx.token, ctx.step, ctx.events = n, token, step, events
        ctx.set_materialize_grads(False)      # an alias nobody consumed contributes None, not a full-size zero tensor to sum
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        if ctx.events:
            # a contribution that was computed on the side stream (the h-gate conv's data gradient, _GateConvLstm.backward): this fan-in is
            # its first reader -- the current stream waits here, after everything that did not depend on it has been enqueued
            cur = torch.cuda.current_stream()
            for ev in ctx.events.values():
                cur.wait_event(ev)
            ctx.events.clear()
        tok = ctx.token if ctx.token is not None else {}
        for i in tok:
            if grads[i] is None:
                raise RuntimeError(f"scanpaths_amd: fan-out alias {i} recorded a split-only gradient but no gradient arrived for it")
        idx = [i for i, g in enumerate(grads) if g is not None]
        if not idx:
            return None, None, None, None, None
        for i in idx:
            if i not in tok and getattr(grads[i], "_sp_skipped", False):
                raise RuntimeError("scanpaths_amd: a gradient whose fp32 form was left unwritten reached a fan-in without its record")
        gs = [grads[i].contiguous() for i in idx]
        ops = [tok.get(i) for i in idx]            # contributions that exist only as split operands (cell backward)
        tok.clear()                                # (a second backward through a retained graph records again)
        if any(op is not None for op in ops):
            out = torch.empty_like(gs[0])
            n = out.numel()
            assert n % 16 == 0 and len(gs) <= 32, (n, len(gs))
            f = (C.c_void_p * len(gs))(*[None if op is not None else g.data_ptr() for g, op in zip(gs, ops)])
            pl = (C.c_void_p * len(gs))(*[op.buf.data_ptr() if op is not None else None for op in ops])
            sc = (C.c_void_p * len(gs))(*[op.scale.data_ptr() if op is not None else None for op in ops])
            hint = _amax_hint(out.device)
            # contributions of decode steps behind a sample's last loss step are exactly zero (rows_ctx): not read
            rcs = [op.rows[0] for op in ops if op is not None and op.rows is not None]
            rc = rcs[0] if (rcs and all(r is rcs[0] for r in rcs) and (n // rcs[0].B) % 16 == 0 and n % rcs[0].B == 0) else None
            steps = (C.c_int * len(gs))(*[(op.rows[1] if (op is not None and op.rows is not None) else -1) for op in ops]) if rc else None
            check(hip.lib().sp_sum_n_mixed_rows(f, pl, sc, len(gs), n, ptr(out), _hint_ptr(hint), ptr(rc.last) if rc else None, steps,
                                                rc.B if rc else 0, hip.stream()), "sp_sum_n_mixed_rows")
            if hint is not None:
                out._sp_amax = hint
            return out, None, None, None, None
        if len(gs) == 1:
            return gs[0], None, None, None, None
        out = torch.empty_like(gs[0])
        n = out.numel()
        if n % 4 or len(gs) > 32:
            acc = gs[0]
            for g in gs[1:]:
                acc = _add_raw(acc, g)
            return acc, None, None, None, None
        arr = (C.c_void_p * len(gs))(*[g.data_ptr() for g in gs])
        hint = _amax_hint(out.device)          # max|sum|: the LSTM cell's backward bounds its split operand with it
        # masked-step sparsity: every term of a tensor that lives at decode step ctx.step (h_t: the two heads' and the next step's
        # h-gate gradients), or the terms a producer marked with their memory update (_sp_rows: semantic pooling -> vf), are exact
        # zeros for the samples behind their last loss step -- not read (a lost hint only makes the pass dense again)
        rc0 = rows_ctx(ctx.step, gs[0].shape[0]) if gs[0].dim() > 1 else None
        marks = [getattr(grads[i], "_sp_rows", None) for i in idx]
        if rc0 is None and any(m is not None for m in marks):
            cand = next(m[0] for m in marks if m is not None)      # (a mark is made by a backward node of THIS pass from its own token)
            if gs[0].dim() > 1 and cand.B == gs[0].shape[0] and ROW_SPARSITY:
                rc0 = cand
        steps = None
        if rc0 is not None and n % rc0.B == 0 and (n // rc0.B) % 4 == 0:
            if ctx.step is not None:
                steps = [int(ctx.step)] * len(gs)
            else:
                steps = [int(m[1]) if (m is not None and m[0] is rc0) else -1 for m in marks]
        if steps is not None and any(st >= 0 for st in steps):
            FUSION_COUNTS["fan_in_rows"] += 1
            check(hip.lib().sp_sum_n_rows(arr, len(gs), n, ptr(out), _hint_ptr(hint), ptr(rc0.last), (C.c_int * len(gs))(*steps), rc0.B,
                                          hip.stream()), "sp_sum_n_rows")
        else:
            check(hip.lib().sp_sum_n(arr, len(gs), n, ptr(out), _hint_ptr(hint), hip.stream()), "sp_sum_n")
        if hint is not None:
            out._sp_amax = hint
        return out, None, None, None, None


def fanout(x: torch.Tensor, n: int, step=None):
    """n aliases of x whose gradients are summed in ONE pass; the operand-split cache and the fused-amax hint travel with them.
    step: x lives at this decode step (dim 0 = samples): under the masked-step sparsity of the backward pass (rows_ctx) the samples behind
    their last loss step contribute exact zeros that the fan-in does not read."""
    split_ok = x.numel() % 16 == 0 and n <= 32
    token = {} if split_ok else None
    events = {} if (ASYNC_DGRAD and step is not None and x.is_cuda) else None
    outs = _FanOut.apply(x, n, token, step, events)
    if events is not None:
        for i, o in enumerate(outs):
            o._sp_fan_ev = (events, i)      # a consumer may compute its gradient on another stream and leave its completion event here
    for attr in ("_sp_cache", "_sp_amax"):
        v = getattr(x, attr, None)
        if v is not None:
            for o in outs:
                setattr(o, attr, v)
    if split_ok:
        for i, o in enumerate(outs):
            o._sp_fan = (token, i)          # a consumer may hand back its gradient as a split operand only (see _FanOut)
            o._sp_split_grad_ok = True
    return outs




class GradMerge:
    """Shared by the two consumers of a ResNet block input (conv1 and the identity branch / the downsample conv): the consumer whose
    backward runs first leaves its input gradient in ``first``; conv1's data-gradient GEMM -- always the last to run, it sits at
    the end of the block's backward chain -- then accumulates into that buffer (epilogue beta = 1) instead of writing its own tensor
    for autograd to add: one read + one write of the block-input gradient instead of a write, two reads and a write."""
    __slots__ = ("first", "merged")

    def __init__(self):
        self.first, self.merged = None, False


class _Tap(Function):
    """x -> (x for conv1, x for the other consumer); backward returns the merged gradient when conv1 accumulated into the other
    consumer's buffer (GradMerge), else the sum of the two."""
    @staticmethod
    def forward(ctx, x, holder):
        ctx.holder = holder
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, g_main, g_side):
        h = ctx.holder
        merged, h.first, h.merged = h.merged, None, False
        if g_main is None or g_side is None:
            return (g_main if g_side is None else g_side), None
        if merged:
            return g_main, None                      # = the other consumer's buffer, conv1's data gradient already added
        return _add_raw(g_main.contiguous(), g_side.contiguous()), None


def tap(x: torch.Tensor, holder: GradMerge):
    outs = _Tap.apply(x, holder)
    for attr in ("_sp_cache", "_sp_amax"):
        v = getattr(x, attr, None)
        if v is not None:
            for o in outs:
                setattr(o, attr, v)
    return outs


class _ScaleConst(Function):
    @staticmethod
    def forward(ctx, x, c):
        x = x.contiguous()
        ctx.c = float(c)
        cs = torch.full((1,), float(c), dtype=torch.float32, device=x.device)
        out = torch.empty_like(x)
        check(hip.lib().sp_scale_by(ptr(x), ptr(cs), x.numel(), ptr(out), hip.stream()), "sp_scale_by")
        return out

    @staticmethod
    def backward(ctx, g):
        return _ScaleConst.apply(g, ctx.c), None


def scale_const(x: torch.Tensor, c: float) -> torch.Tensor:
    """x * c (python constant), differentiable"""
    return _ScaleConst.apply(x, c)


def add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a + b (same shape) as a differentiable HIP op."""
    return _Add.apply(a, b)


def _phys(w: torch.Tensor) -> torch.Tensor:
    """[Co,KH,KW,Ci] contiguous view (or copy) of a logical OIHW weight."""
    p = w.permute(0, 2, 3, 1)
    return p if p.is_contiguous() else p.contiguous()


def _out_hw(H, W, KH, KW, stride, pad, dil):
    return (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1, (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1


# ----------------------------------------------------------------------------------------------------
# convolution
# ----------------------------------------------------------------------------------------------------
class _Conv2d(Function):
    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, dil, relu, wcache=None, bn_stats=False, grad_store=None, grad_accum=None, step=None):
        # grad_store / grad_accum: GradMerge of a block input -- leave the input gradient there / accumulate into what is there
        # step: the decode step this application belongs to (row sparsity of its backward, see rows_ctx)
        ctx.grad_store, ctx.grad_accum, ctx.step = grad_store, grad_accum, step
        ctx.w_param = _grad_slot(w) if w.dim() == 4 else None      # its gradient may go straight into the flat buffer
        defer = wcache.get("defer") if isinstance(wcache, dict) else None       # DeferredWgrad of a weight applied T times
        ctx.defer_final = defer.claim() if (defer is not None and ctx.needs_input_grad[1]) else False
        ctx.dy_token = None
        # wcache: dict shared by all applications of the SAME weight inside one forward/backward (the h-gate conv runs T times):
        # its split forms ("w": forward operand, "wT": data-gradient operand) are produced once instead of per step
        x = x.contiguous()
        N, H, W_, Ci = x.shape
        wp = _phys(w.detach())
        Co, KH, KW, Ciw = wp.shape
        assert Ciw == Ci, (wp.shape, x.shape)
        Ho, Wo = _out_hw(H, W_, KH, KW, stride, pad, dil)
        y = torch.empty((N, Ho, Wo, Co), dtype=torch.float32, device=x.device)
        xs = None
        shared = getattr(x, "_sp_cache", None) is not None
        if getattr(x, "_sp_uninit", False) and not conv_runs_from_split(x.shape, w, stride, pad, dil, ctx.needs_input_grad[1]):
            raise RuntimeError("scanpaths_amd: a BatchNorm left this input's fp32 form unwritten (skip_z) but this conv does not run "
                               "from the split operand alone; run with SP_BN_SKIP_DX=0")
        if _b3_pays(N * Ho * Wo, Co, KH * KW * Ci, Ci, a_elems=x.numel(), free_a=shared):
            xs = split_op(x, channel=True)
            wsplit = _weight_operand(wp, xs, wcache)
            stats = None
            if bn_stats and BN_SPLIT and xs.scheme == "f16x2" and not THROUGHPUT_MODE and bias is None and not relu:
                # bn_stats: a train-mode BatchNorm follows -- the epilogue writes the first stage of its batch statistics
                dd = ConvDesc(N, H, W_, Ci, Ci, Ho, Wo, Co, Co, KH, KW, stride, pad, dil, 0, KH * KW * Ci, 1.0, 0, 0, 1, 0, 0, 0, 0, None)
                G = hip.lib().sp_conv_stats_tiles(C.byref(dd))         # M-tiles of the kernel that will run this conv
                stats = (torch.empty((G, 2, Co), dtype=torch.float64, device=x.device),
                         torch.empty((G, 2, Co), dtype=torch.float32, device=x.device), G)
            _igemm_b3(xs, wsplit, bias, y, N_img=N, Hi=H, Wi=W_, Kc=Ci, ldx=Ci, Ho=Ho, Wo=Wo, Nout=Co, ldc=Co,
                      ldw=KH * KW * Ci, KH=KH, KW=KW, stride=stride, pad=pad, dil=dil, mode=0, relu=relu, stats=stats)
            y._sp_from_split = xs.scheme == "f16x2"        # a BatchNorm behind this conv may emit the split gradient (bn_act)
            if stats is not None:
                FUSION_COUNTS["conv_bn_stats"] += 1
                y._sp_bnstats = stats
            # Both backward GEMMs of this conv would read ONLY the split form of the output gradient: a BatchNorm behind it (the
            # single consumer of y, see bn_act skip_dx) may then leave the fp32 gradient unwritten.  The token lets this conv's
            # backward fail loudly if such a gradient ever arrives without its split form.
            wsch = _wgrad_scheme(Ci, Co)
            if (y._sp_from_split and wsch == "f16x2" and bias is None and not relu
                    and (not ctx.needs_input_grad[1] or (xs.scheme == wsch and _w3_pays(N * Ho * Wo, Co, KH * KW * Ci, Ci,
                                                                                          free_splits=True)))
                    and (not ctx.needs_input_grad[0] or _b3_pays(N * H * W_, Ci, KH * KW * Co, Co, a_elems=y.numel(),
                                                                  free_a=True))):
                ctx.dy_token = y._sp_dy_token = {"skipped": False}
            # the weight-gradient GEMM consumes the same split operand: keep it (6 B/element) instead of re-splitting x in
            # backward (HBM pass of 10 B/element per conv); sized for 288 GB
            if not (ctx.needs_input_grad[1] and _w3_pays(N * Ho * Wo, Co, KH * KW * Ci, Ci, free_splits=True)
                    and xs.scheme == _wgrad_scheme(Ci, Co)):
                xs = None
        else:
            _igemm(x, wp, bias, y, N_img=N, Hi=H, Wi=W_, Kc=Ci, ldx=Ci, Ho=Ho, Wo=Wo, Nout=Co, ldc=Co, ldw=KH * KW * Ci,
                   KH=KH, KW=KW, stride=stride, pad=pad, dil=dil, mode=0, relu=relu)
        ctx.xs_scheme = (xs.scheme, xs.kind) if xs is not None else None
        xs_buf, xs_scale = (xs.buf, xs.scale) if xs is not None else (None, None)
        ctx.cfg = (stride, pad, dil, relu, bias is not None)
        ctx.wcache = wcache
        ctx.save_for_backward(x, wp, y if relu else None, xs_buf, xs_scale)
        return y

    @staticmethod
    def backward(ctx, dy):
        stride, pad, dil, relu, has_bias = ctx.cfg
        x, wp, y, xs_buf, xs_scale = ctx.saved_tensors
        xs = SplitOperand(xs_buf, xs_scale, *ctx.xs_scheme) if xs_buf is not None else None
        if ctx.dy_token is not None and ctx.dy_token["skipped"] and getattr(dy, "_sp_cache", None) is None:
            raise RuntimeError("scanpaths_amd: a BatchNorm left this gradient's fp32 form unwritten (skip_dx) but its split form "
                               "did not arrive with it; run with SP_BN_SKIP_DX=0")
        dy = dy.contiguous()
        if relu:
            dyr = torch.empty_like(dy)
            check(hip.lib().sp_relu_bwd(ptr(dy), ptr(y), dy.numel(), ptr(dyr), hip.stream()), "sp_relu_bwd")
            dy = dyr
        dx, dw = _conv_backward(x, wp, dy, xs, stride, pad, dil, ctx.wcache, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                accum=ctx.grad_accum, defer_final=ctx.defer_final, step=ctx.step, w_param=ctx.w_param)
        if ctx.grad_store is not None and dx is not None:
            ctx.grad_store.first = dx
        db = _colsum_any(dy, wp.shape[0]) if (has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db, None, None, None, None, None, None, None, None, None




class DeferredWgrad:
    """Weight gradient of a conv that is applied T times with the same weight (the h-gate conv of the ConvLSTM): nobody needs the T
    contributions before the end of backpropagation through time, and the split operands of every application (h_{t-1}: kept by the
    forward pass; the gate gradient of step t: written by the cell backward) stay alive until then anyway.  Each application's
    backward only RECORDS its operand pair; the application whose backward runs LAST (the first one of the forward pass claims that
    role) issues ONE launch over all of them (sp_conv_wgrad_f16x2_multi: hw2_kernel, 256 x 256 tiles; the per-step launches ran 4.5
    rounds of 256 workgroups each and wrote / re-read 8 slabs per step) and hands the sum to autograd; the others return no gradient.
    (Round 2's side-stream variant of this class measured neutral -- DESIGN.md 5f -- and is gone.)"""
    __slots__ = ("items", "claimed")

    def __init__(self):
        self.items, self.claimed = [], False

    def claim(self) -> bool:
        first, self.claimed = not self.claimed, True
        return first


def _flush_deferred(defer, wp, geom):
    """dW of all recorded applications: one multi-application launch where the kernel's shape constraints hold, else one launch per
    application accumulating in place"""
    items, defer.items, defer.claimed = defer.items, [], False
    if all(st is not None for _, _, st in items):
        # decode-step order, earliest first (backward recorded them latest first).  The launch dispatches its workgroups segment by
        # segment: under the masked-step sparsity the early steps have the most live samples, i.e. the longest workgroups -- started
        # first they leave a tail of short ones (latest-first ended the launch on its longest workgroups).  Dense and sparse launches
        # use the same order, so the fixed-order slab reduce keeps them bit-identical.
        items = sorted(items, key=lambda it: it[2])
    Co, KH, KW, Ci = wp.shape
    dwp = torch.empty_like(wp)
    L = hip.lib()
    xs0, dys0, _ = items[0]
    same = all(x.kind == xs0.kind and y.kind == dys0.kind for x, y, _ in items)
    steps = [st for _, _, st in items]
    toks = [getattr(st, "rows", None) for st in steps]
    rc = rows_ctx(steps[0], geom["N_img"]) if (toks and toks[0] is not None and all(tk is toks[0] for tk in toks)) else None
    d = WgradDesc(geom["N_img"], geom["Hi"], geom["Wi"], Ci, Ci, geom["Ho"], geom["Wo"], Co, geom.get("ldy", Co), KH, KW, geom["stride"],
                  geom["pad"], geom["dil"], KH * KW * Ci, 0, 1.0, 1, 0, 0, 0)
    d.x_scale_vec, d.y_scale_vec = int(xs0.kind == "cols"), int(dys0.kind == "cols")
    nseg = len(items)
    wsb = L.sp_conv_wgrad_f16x2_multi_workspace(C.byref(d), nseg) if (same and not THROUGHPUT_MODE and 1 < nseg <= 16) else 0
    if wsb > 0:
        ws = hip.workspace(wsb, wp.device, slot=3)
        arr = lambda ts: (C.c_void_p * nseg)(*[t.data_ptr() for t in ts])
        Xa, Sxa = arr([x.buf for x, _, _ in items]), arr([x.scale for x, _, _ in items])
        Ya, Sya = arr([y.buf for _, y, _ in items]), arr([y.scale for _, y, _ in items])
        seg_steps = (C.c_int * nseg)(*steps) if rc is not None else None
        if rc is not None:
            FUSION_COUNTS["row_sparse_bwd"] += 1

        def launch():
            check(L.sp_conv_wgrad_f16x2_multi(C.byref(d), nseg, Xa, Sxa, Ya, Sya, ptr(dwp), ptr(ws), ptr(rc.last) if rc is not None else None,
                                              seg_steps, hip.stream()), "sp_conv_wgrad_f16x2_multi")
        FUSION_COUNTS["wgrad_multi"] += 1
        if hip.TIMER is None:
            launch()
        else:
            M = geom["N_img"] * geom["Ho"] * geom["Wo"]
            hip.TIMER.bracket(("h2_wgrad_multi" + ("_rows" if rc is not None else ""), M * nseg, Co, KH * KW * Ci, f"{KH}x{KW}", nseg),
                              2.0 * M * nseg * Co * KH * KW * Ci, launch)
    else:
        g2 = {k: v for k, v in geom.items() if k != "ldy"}
        for k, (x, y, _) in enumerate(items):
            _wgrad_b3(x, y, dwp, beta=int(k > 0), ldo=KH * KW * Ci, Ci=Ci, Co=Co, KH=KH, KW=KW, **g2)
    return dwp.permute(0, 3, 1, 2)


def _conv_backward(x, wp, dy, xs, stride, pad, dil, wcache, need_dx, need_dw, accum=None, defer_final=False, step=None, w_param=None):
    """data and weight gradient of y = conv(x, wp) (NHWC, physical weight [Co,KH,KW,Ci]); xs: the forward's split x or None;
    accum: GradMerge whose ``first`` (another consumer's gradient of x) the data gradient is added to in place"""
    N, H, W_, Ci = x.shape
    Co, KH, KW, _ = wp.shape
    _, Ho, Wo, _ = dy.shape
    dx = dw = None
    dys = None
    dy_cached = getattr(dy, "_sp_cache", None)       # the producer of dy (a BatchNorm / cell backward) already wrote its split form
    defer = wcache.get("defer") if (DEFER_WGRAD and isinstance(wcache, dict)) else None
    geom = dict(N_img=N, Hi=H, Wi=W_, Ho=Ho, Wo=Wo, stride=stride, pad=pad, dil=dil)
    rc = rows_ctx(step, N)          # samples whose gradient rows are exactly zero at this decode step (loss masks)
    deferred = False
    if need_dw and defer is not None:
        wsch = _wgrad_scheme(Ci, Co)
        if wsch == "f16x2" and xs is not None and xs.scheme == wsch and \
                _w3_pays(N * Ho * Wo, Co, KH * KW * Ci, Ci, free_splits=dy_cached is not None and wsch in dy_cached):
            dys = dy_cached[wsch] if (dy_cached is not None and wsch in dy_cached) else split_op(dy, wsch, channel=True)
            defer.items.append((xs, dys, step))
            deferred = True
    if need_dx:
        beta = 0
        if accum is not None and accum.first is not None and accum.first.shape == x.shape and accum.first.is_contiguous():
            dx, beta, accum.merged = accum.first, 1, True
            FUSION_COUNTS["grad_merge"] += 1
        else:
            dx = torch.empty_like(x)
        if _b3_pays(N * H * W_, Ci, KH * KW * Co, Co, a_elems=dy.numel(), free_a=dy_cached is not None or dys is not None):
            if dys is None or dys.scheme != _scheme_for(Co):
                dys = split_op(dy, channel=True)
            wT = _weight_operand(wp, dys, wcache, transposed=True)
            _igemm_b3(dys, wT, None, dx, N_img=N, Hi=Ho, Wi=Wo, Kc=Co, ldx=Co, Ho=H, Wo=W_, Nout=Ci,
                      ldc=Ci, ldw=KH * KW * Co, KH=KH, KW=KW, stride=stride, pad=pad, dil=dil, mode=1, beta=beta,
                      rows=(rc, step) if rc is not None else None)
        else:
            _fp32_required(dy, "the fp32 data-gradient GEMM")
            _igemm(dy, wp, None, dx, N_img=N, Hi=Ho, Wi=Wo, Kc=Co, ldx=Co, Ho=H, Wo=W_, Nout=Ci, ldc=Ci, ldw=Ci, KH=KH,
                   KW=KW, stride=stride, pad=pad, dil=dil, mode=1, beta=beta)
    slot = None
    if need_dw and not deferred:
        if defer is None:          # (a weight with deferred applications is summed by _flush_deferred: ordinary path)
            slot, dwp = _take_grad_view(w_param, (0, 2, 3, 1))
        if slot is None:
            dwp = torch.empty_like(wp)
        wsch = _wgrad_scheme(Ci, Co)
        if dys is None and dy_cached is not None and wsch in dy_cached:
            dys = dy_cached[wsch]
        free = xs is not None and (dys is not None and dys.scheme == wsch)
        if _w3_pays(N * Ho * Wo, Co, KH * KW * Ci, Ci, free_splits=free):
            _wgrad_b3(xs if xs is not None else split_op(x, wsch, channel=True),
                      dys if dys is not None and dys.scheme == wsch else split_op(dy, wsch, channel=True), dwp, N_img=N, Hi=H,
                      Wi=W_, Ci=Ci, Ho=Ho, Wo=Wo, Co=Co, ldo=KH * KW * Ci, KH=KH, KW=KW, stride=stride, pad=pad, dil=dil)
        else:
            _fp32_required(dy, "the fp32 weight-gradient GEMM")
            _wgrad(x, dy, dwp, N_img=N, Hi=H, Wi=W_, Ci=Ci, ldx=Ci, Ho=Ho, Wo=Wo, Co=Co, ldy=Co, ldo=KH * KW * Ci,
                   KH=KH, KW=KW, stride=stride, pad=pad, dil=dil)
        if slot is not None:
            slot.done()            # written into the parameter's flat-buffer view: autograd gets None
        else:
            dw = dwp.permute(0, 3, 1, 2)
    if need_dw and defer is not None and defer_final and defer.items:
        # this application runs last in backward: the recorded applications (its own among them, unless it took the plain path) in
        # one launch
        dsum = _flush_deferred(defer, wp, geom)
        dw = dsum if dw is None else dw + dsum
    return dx, dw


def conv_takes_split(x_shape, w, stride=1, pad=0, dil=1) -> bool:
    """would conv2d(x, w) run on the 2xfp16 split path if x [N,H,W,Ci] arrived with its split operand attached (bn_act emit_split)"""
    N, H, W_, Ci = x_shape
    Co, _, KH, KW = w.shape
    Ho, Wo = _out_hw(H, W_, KH, KW, stride, pad, dil)
    return _scheme_for(Ci) == "f16x2" and _b3_pays(N * Ho * Wo, Co, KH * KW * Ci, Ci, a_elems=N * H * W_ * Ci, free_a=True)


def conv_runs_from_split(x_shape, w, stride=1, pad=0, dil=1, need_dw=True) -> bool:
    """conv2d(x, w) reads x only through its split operand: forward on the split path, and (if the weight needs a gradient) the
    weight-gradient GEMM from the split operand kept by the forward pass"""
    N, H, W_, Ci = x_shape
    Co, _, KH, KW = w.shape
    Ho, Wo = _out_hw(H, W_, KH, KW, stride, pad, dil)
    if not conv_takes_split(x_shape, w, stride, pad, dil):
        return False
    return (not need_dw) or (_wgrad_scheme(Ci, Co) == "f16x2" and _w3_pays(N * Ho * Wo, Co, KH * KW * Ci, Ci, free_splits=True))


def conv2d(x, w, bias=None, stride=1, pad=0, dil=1, relu=False, wcache=None, bn_stats=False, grad_store=None, grad_accum=None, step=None):
    return _Conv2d.apply(x, w, bias, stride, pad, dil, relu, wcache, bn_stats, grad_store, grad_accum, step)


class _PadLast(Function):
    """[..., Cin] -> [..., Cout] zero padded (stem weight 3->4 channels, K padding of tiny GEMM operands)."""
    @staticmethod
    def forward(ctx, x, cout):
        x = x.contiguous()
        cin = x.shape[-1]
        rows = x.numel() // cin
        y = torch.empty(x.shape[:-1] + (cout,), dtype=torch.float32, device=x.device)
        check(hip.lib().sp_pad_lastdim(ptr(x), rows, cin, cout, ptr(y), hip.stream()), "sp_pad_lastdim")
        ctx.cin = cin
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        cout = dy.shape[-1]
        rows = dy.numel() // cout
        dx = torch.empty(dy.shape[:-1] + (ctx.cin,), dtype=torch.float32, device=dy.device)
        check(hip.lib().sp_pad_lastdim(ptr(dy), rows, cout, ctx.cin, ptr(dx), hip.stream()), "sp_pad_lastdim")
        return dx, None


def pad_last(x, cout):
    return _PadLast.apply(x, cout)


def nchw_to_nhwc(x: torch.Tensor, cpad: int) -> torch.Tensor:
    """images [N,C,H,W] -> [N,H,W,cpad]; inputs carry no gradient."""
    x = x.contiguous()
    N, Cc, H, W_ = x.shape
    y = torch.empty((N, H, W_, cpad), dtype=torch.float32, device=x.device)
    check(hip.lib().sp_nchw_to_nhwc_pad(ptr(x), N, Cc, H, W_, cpad, ptr(y), hip.stream()), "sp_nchw_to_nhwc_pad")
    return y


# ----------------------------------------------------------------------------------------------------
# dense / batched GEMM  (nn.Linear, weight composition, rank-1 contraction, pooled features)
# ----------------------------------------------------------------------------------------------------
def _rows(t):   # (nbatch, M, K, ld, batch_stride) of a 2-D / 3-D operand whose last dim is contiguous
    assert t.stride(-1) == 1
    if t.dim() == 2:
        return 1, t.shape[0], t.shape[1], t.stride(0), 0
    assert t.dim() == 3
    return t.shape[0], t.shape[1], t.shape[2], t.stride(1), t.stride(0)


class DeferredGemmWgrad:
    """Weight gradient of a dense layer that is applied T times with the same weight (spatial_embed / semantic_embed and the contracted
    filters of the rank-1 gate terms: once per memory update / decode step): dW = sum_t dC_t^T A_t is ONE GEMM over the concatenated
    rows, K = T x rows, at the end of backward instead of T GEMMs with K = rows (32 .. 64) that each write the whole weight-sized
    gradient (26 .. 56 MB) for a fan-in pass to re-read.  Each application's backward records (A_t, dC_t); the first application of
    the forward pass -- the last to run in backward -- issues the GEMM, the others return no gradient (pass the weight itself to every
    application, not fan-out aliases)."""
    __slots__ = ("items", "claimed")

    def __init__(self):
        self.items, self.claimed = [], False

    def claim(self) -> bool:
        first, self.claimed = not self.claimed, True
        return first


class _Gemm(Function):
    """C = relu?(alpha * A @ op(B) + bias).  layout 'nk': B is [N,K] (C = A B^T); 'kn': B is [K,N].
    A: [M,K] or [Bt,M,K]; B 2-D (shared) or 3-D (batched).  defer: DeferredGemmWgrad of a weight applied several times ('nk' only)."""
    @staticmethod
    def forward(ctx, a, b, bias, layout, alpha, relu, defer=None):
        ctx.defer, ctx.defer_final = None, False
        if defer is not None and ctx.needs_input_grad[1]:
            assert layout == "nk" and (a.dim() == b.dim() or (a.dim() == 2 and b.dim() == 2)), (layout, a.shape, b.shape)
            ctx.defer, ctx.defer_final = defer, defer.claim()
        a = a if a.stride(-1) == 1 else a.contiguous()
        b = b if b.is_contiguous() else b.contiguous()
        nb, M, K, lda, sA = _rows(a)
        bb = b.dim() == 3
        if layout == "nk":
            Nn, Kb = b.shape[-2], b.shape[-1]
        else:
            Kb, Nn = b.shape[-2], b.shape[-1]
        assert Kb == K, (a.shape, b.shape, layout)
        if bb:
            assert b.shape[0] == nb
        out_shape = (M, Nn) if a.dim() == 2 else (nb, M, Nn)
        c = torch.empty(out_shape, dtype=torch.float32, device=a.device)
        sW = b.stride(0) if bb else 0
        _igemm(a, b, bias, c, N_img=M, Hi=1, Wi=1, Kc=K, ldx=lda, Ho=1, Wo=1, Nout=Nn, ldc=Nn,
               ldw=(K if layout == "nk" else Nn), mode=(0 if layout == "nk" else 1), alpha=alpha, relu=relu, nbatch=nb,
               sX=sA, sW=sW, sC=M * Nn)
        ctx.cfg = (layout, alpha, relu, bias is not None)
        ctx.save_for_backward(a, b, c if relu else None)
        return c

    @staticmethod
    def backward(ctx, dc):
        layout, alpha, relu, has_bias = ctx.cfg
        a, b, c = ctx.saved_tensors
        dc = dc.contiguous()
        L = hip.lib()
        if relu:
            t = torch.empty_like(dc)
            check(L.sp_relu_bwd(ptr(dc), ptr(c), dc.numel(), ptr(t), hip.stream()), "sp_relu_bwd")
            dc = t
        nb, M, K, lda, sA = _rows(a)
        Nn = dc.shape[-1]
        bb = b.dim() == 3
        sW = b.stride(0) if bb else 0
        da = db = dbias = None
        if ctx.needs_input_grad[0]:
            da = torch.empty(a.shape, dtype=torch.float32, device=a.device)
            # dA[M,K] = alpha * dC[M,N] @ (B as [N,K] rows)   -> 'kn' on an [N][K] matrix, or 'nk' on a [K][N] one
            if layout == "nk":
                _igemm(dc, b, None, da, N_img=M, Hi=1, Wi=1, Kc=Nn, ldx=Nn, Ho=1, Wo=1, Nout=K, ldc=K, ldw=K, mode=1,
                       alpha=alpha, nbatch=nb, sX=M * Nn, sW=sW, sC=M * K)
            else:
                _igemm(dc, b, None, da, N_img=M, Hi=1, Wi=1, Kc=Nn, ldx=Nn, Ho=1, Wo=1, Nout=K, ldc=K, ldw=Nn, mode=0,
                       alpha=alpha, nbatch=nb, sX=M * Nn, sW=sW, sC=M * K)
        if ctx.needs_input_grad[1] and ctx.defer is not None:
            ctx.defer.items.append((a, dc))
            if ctx.defer_final:          # this application runs last in backward: one GEMM over the rows of all recorded applications
                items, ctx.defer.items, ctx.defer.claimed = ctx.defer.items, [], False
                ac = torch.cat([x for x, _ in items], -2) if len(items) > 1 else items[0][0]
                dcc = torch.cat([y for _, y in items], -2) if len(items) > 1 else items[0][1]
                ac = ac if ac.stride(-1) == 1 else ac.contiguous()
                nbc, Mc, _, ldac, sAc = _rows(ac)
                db = torch.empty(b.shape, dtype=torch.float32, device=b.device)
                _wgrad(ac, dcc, db, N_img=Mc, Hi=1, Wi=1, Ci=K, ldx=ldac, Ho=1, Wo=1, Co=Nn, ldy=Nn, ldo=K, alpha=alpha,
                       nbatch=nbc, sX=sAc, sY=Mc * Nn, sO=(b.stride(0) if bb else 0))
        elif ctx.needs_input_grad[1]:
            db = torch.empty(b.shape, dtype=torch.float32, device=b.device)
            if bb or nb == 1:
                rows, nbt, sAa, sCc, sO = M, nb, sA, M * Nn, (b.stride(0) if bb else 0)
            else:   # shared B, batched A: one reduction over all Bt*M rows (A must be densely stacked)
                assert sA == M * lda, "batched A with shared B must be contiguous over the batch"
                rows, nbt, sAa, sCc, sO = nb * M, 1, 0, 0, 0
            if layout == "nk":    # dB[N,K] = alpha * dC^T A
                _wgrad(a, dc, db, N_img=rows, Hi=1, Wi=1, Ci=K, ldx=lda, Ho=1, Wo=1, Co=Nn, ldy=Nn, ldo=K, alpha=alpha,
                       nbatch=nbt, sX=sAa, sY=sCc, sO=sO)
            else:                 # dB[K,N] = alpha * A^T dC
                _wgrad(dc, a, db, N_img=rows, Hi=1, Wi=1, Ci=Nn, ldx=Nn, Ho=1, Wo=1, Co=K, ldy=lda, ldo=Nn, alpha=alpha,
                       nbatch=nbt, sX=sCc, sY=sAa, sO=sO)
        if has_bias and ctx.needs_input_grad[2]:
            dbias = _colsum_any(dc, Nn)
        return da, db, dbias, None, None, None, None


def gemm(a, b, bias=None, layout="nk", alpha=1.0, relu=False, defer=None):
    return _Gemm.apply(a, b, bias, layout, alpha, relu, defer)


def linear(x, weight, bias, defer=None):
    """nn.Linear on the last dim (weight [out,in]); defer: DeferredGemmWgrad shared by the applications of this weight."""
    shp = x.shape
    y = gemm(x.reshape(-1, shp[-1]), weight, bias, "nk", defer=defer)
    return y.view(*shp[:-1], weight.shape[0])


# ----------------------------------------------------------------------------------------------------
# BatchNorm (+residual, +ReLU), maxpool
# ----------------------------------------------------------------------------------------------------
class _BnAct(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, rmean, rvar, residual, training, momentum, eps, relu):
        x = x.contiguous()
        Cc = x.shape[-1]
        M = x.numel() // Cc
        L = hip.lib()
        dev = x.device
        mean = torch.empty(Cc, dtype=torch.float32, device=dev)
        invstd = torch.empty(Cc, dtype=torch.float32, device=dev)
        if training:
            ws = hip.workspace(L.sp_bn_workspace(M, Cc), dev, slot=1)
            check(L.sp_bn_stats(ptr(x), M, Cc, eps, momentum, ptr(mean), ptr(invstd), ptr(rmean), ptr(rvar), ptr(ws),
                                hip.stream()), "sp_bn_stats")
        else:
            check(L.sp_bn_eval_stats(ptr(rmean), ptr(rvar), Cc, eps, ptr(mean), ptr(invstd), hip.stream()),
                  "sp_bn_eval_stats")
        y = torch.empty_like(x)
        res = residual.contiguous() if residual is not None else None
        hint = _amax_hint(dev)
        check(L.sp_bn_apply(ptr(x), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), ptr(res), int(relu), M, Cc, ptr(y),
                            _hint_ptr(hint), hip.stream()), "sp_bn_apply")
        if hint is not None:
            y._sp_amax = hint
        ctx.cfg = (training, relu, residual is not None)
        ctx.save_for_backward(x, y if relu else None, mean, invstd, gamma.detach())
        return y

    @staticmethod
    def backward(ctx, dy):
        training, relu, has_res = ctx.cfg
        x, y, mean, invstd, gamma = ctx.saved_tensors
        dy = dy.contiguous()
        Cc = x.shape[-1]
        M = x.numel() // Cc
        L = hip.lib()
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if has_res else None
        dgamma = torch.empty(Cc, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(Cc, dtype=torch.float32, device=x.device)
        ws = hip.workspace(L.sp_bn_workspace(M, Cc), x.device, slot=1)
        hint = _amax_hint(x.device)
        check(L.sp_bn_backward(ptr(dy), ptr(x), ptr(y), ptr(mean), ptr(invstd), ptr(gamma), int(relu), int(training), M, Cc,
                               ptr(dx), ptr(dres), ptr(dgamma), ptr(dbeta), ptr(ws), _hint_ptr(hint), hip.stream()),
              "sp_bn_backward")
        if hint is not None:
            dx._sp_amax = hint
        return dx, dgamma, dbeta, None, None, dres, None, None, None, None




class _BnActSplit(Function):
    """Train-mode BatchNorm(+residual, +ReLU) whose passes also emit what their consumers need (sp_bn_fwd_split / sp_bn_bwd_split):
    the 2xfp16 split operand of the consumer conv (forward: of the output; backward: of the gradient w.r.t. the conv output, both
    GEMMs of the producing conv's backward read it), with the operand scale taken from an upper bound of the maximum instead of a
    measured one, and the ReLU mask as one bit per element instead of the fp32 output.  emit_fwd / emit_bwd: whether the split
    operands will be read (else only the bound is attached as the max|.| hint)."""
    @staticmethod
    def forward(ctx, x, gamma, beta, rmean, rvar, residual, momentum, eps, relu, emit_fwd, emit_bwd, pre, res_store=None,
                dy_token=None, skip_z=False):
        ctx.res_store = res_store          # GradMerge: the residual's gradient is left there for conv1's data gradient to add to
        ctx.params = (_grad_slot(gamma), _grad_slot(beta))
        ctx.dy_token = dy_token            # set: the producing conv's backward reads only the split gradient -> dx stays unwritten
        # pre: (partial, mm, G) -- first statistics stage already done by the producing conv's epilogue (conv2d bn_stats=True)
        x = x.contiguous()
        Cc = x.shape[-1]
        M = x.numel() // Cc
        n = x.numel()
        L = hip.lib()
        dev = x.device
        mean = torch.empty(Cc, dtype=torch.float32, device=dev)
        invstd = torch.empty(Cc, dtype=torch.float32, device=dev)
        ext = torch.empty(2 * Cc, dtype=torch.float32, device=dev)
        ws = hip.workspace(L.sp_bn_split_workspace(M, Cc), dev, slot=1)
        emit_fwd = bool(emit_fwd) and Cc % 16 == 0
        planes = torch.empty(2 * n + 32, dtype=torch.float16, device=dev) if emit_fwd else None
        mask = torch.empty(L.sp_bn_mask_words(M, Cc), dtype=torch.int64, device=dev) if relu else None
        y = torch.empty_like(x)
        res = residual.contiguous() if residual is not None else None
        zhint, bhint = _amax_hint(dev), _amax_hint(dev)
        skip_z = bool(skip_z) and planes is not None          # the only consumer reads the split operand: fp32 output unwritten
        check(L.sp_bn_fwd_split(ptr(x), M, Cc, eps, momentum, ptr(gamma), ptr(beta), ptr(res),
                                _hint_ptr(residual._sp_amax) if residual is not None else None, int(relu), ptr(mean), ptr(invstd),
                                ptr(rmean), ptr(rvar), ptr(ext), None if skip_z else ptr(y), ptr(planes), ptr(zhint),
                                _hint_ptr(bhint), ptr(mask),
                                ptr(ws), ptr(pre[0]) if pre else None, ptr(pre[1]) if pre else None, pre[2] if pre else 0,
                                hip.stream()), "sp_bn_fwd_split")
        y._sp_amax = zhint
        FUSION_COUNTS["bn_fwd_split"] += 1
        if planes is not None:
            FUSION_COUNTS["bn_fwd_split_operand"] += 1
            y._sp_cache = {"f16x2": SplitOperand(planes, zhint, "f16x2")}
        if skip_z:
            FUSION_COUNTS["bn_skip_z"] += 1
            y._sp_uninit = True            # conv2d refuses such an input unless it runs (forward and backward) from the split operand
        ctx.cfg = (relu, residual is not None, bool(emit_bwd) and Cc % 16 == 0)
        ctx.save_for_backward(x, mask, mean, invstd, gamma.detach(), ext)
        return y

    @staticmethod
    def backward(ctx, dy):
        relu, has_res, emit = ctx.cfg
        x, mask, mean, invstd, gamma, ext = ctx.saved_tensors
        dy = dy.contiguous()
        Cc = x.shape[-1]
        M = x.numel() // Cc
        L = hip.lib()
        dev = x.device
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if has_res else None
        gslot, dgamma = _take_grad_view(ctx.params[0]) if ctx.needs_input_grad[1] else (None, None)
        bslot, dbeta = _take_grad_view(ctx.params[1]) if ctx.needs_input_grad[2] else (None, None)
        if dgamma is None:
            dgamma = torch.empty(Cc, dtype=torch.float32, device=dev)
        if dbeta is None:
            dbeta = torch.empty(Cc, dtype=torch.float32, device=dev)
        planes = torch.empty(2 * x.numel() + 32, dtype=torch.float16, device=dev) if emit else None
        ws = hip.workspace(L.sp_bn_split_workspace(M, Cc), dev, slot=1)
        dhint, bhint = _amax_hint(dev), _amax_hint(dev)
        skip = emit and ctx.dy_token is not None          # the fp32 dx is allocated (autograd wants a tensor) but never written
        FUSION_COUNTS["bn_bwd_split"] += 1
        if emit:
            FUSION_COUNTS["bn_bwd_split_operand"] += 1
        if skip:
            FUSION_COUNTS["bn_skip_dx"] += 1
            ctx.dy_token["skipped"] = True
        check(L.sp_bn_bwd_split(ptr(dy), ptr(x), ptr(mask), ptr(mean), ptr(invstd), ptr(gamma), ptr(ext), M, Cc,
                                None if skip else ptr(dx),
                                ptr(dres), ptr(planes), ptr(dhint), _hint_ptr(bhint), ptr(dgamma), ptr(dbeta), ptr(ws),
                                hip.stream()), "sp_bn_bwd_split")
        dx._sp_amax = dhint
        if planes is not None:
            dx._sp_cache = {"f16x2": SplitOperand(planes, dhint, "f16x2")}
        if ctx.res_store is not None and dres is not None:
            ctx.res_store.first = dres
        if gslot is not None:
            gslot.done()
            dgamma = None
        if bslot is not None:
            bslot.done()
            dbeta = None
        return dx, dgamma, dbeta, None, None, dres, None, None, None, None, None, None, None, None, None




def bn_act(x, gamma, beta, rmean, rvar, residual=None, training=True, momentum=0.1, eps=1e-5, relu=True, emit_split=False,
           res_store=None, skip_dx=False, skip_z=False):
    """emit_split: the output feeds a conv that runs on the 2xfp16 split path -- the BatchNorm pass writes that operand itself.
    skip_dx: x is a conv output with NO other consumer; when that conv's backward reads only the split gradient (it says so with a
    token on x) the backward pass here writes the split gradient alone and leaves the fp32 one unwritten.
    skip_z: the output's ONLY consumer is a conv that runs forward and backward from the split operand (conv_runs_from_split): the
    fp32 output is allocated but not written.  Neither tensor may be handed to code outside the encoder."""
    if (training and BN_SPLIT and x.shape[-1] % 4 == 0 and _amax_hint_active()
            and (residual is None or getattr(residual, "_sp_amax", None) is not None)):
        emit_bwd = getattr(x, "_sp_from_split", False)       # the producing conv's backward GEMMs read the split gradient
        token = getattr(x, "_sp_dy_token", None) if (skip_dx and BN_SKIP_DX and emit_bwd) else None
        return _BnActSplit.apply(x, gamma, beta, rmean, rvar, residual, momentum, eps, relu, emit_split, emit_bwd,
                                 getattr(x, "_sp_bnstats", None), res_store, token, skip_z and emit_split and BN_SKIP_DX and BN_SKIP_Z)
    return _BnAct.apply(x, gamma, beta, rmean, rvar, residual, training, momentum, eps, relu)


class _MaxPool(Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        N, H, W_, Cc = x.shape
        Ho = -(-(H - 3) // 2) + 1
        Wo = -(-(W_ - 3) // 2) + 1
        if (Ho - 1) * 2 >= H:
            Ho -= 1
        if (Wo - 1) * 2 >= W_:
            Wo -= 1
        y = torch.empty((N, Ho, Wo, Cc), dtype=torch.float32, device=x.device)
        # the window position of each maximum (1 byte per output) replaces x and y in the backward pass
        amx = torch.empty((N, Ho, Wo, Cc), dtype=torch.uint8, device=x.device) if ctx.needs_input_grad[0] else None
        check(hip.lib().sp_maxpool3s2_fwd_idx(ptr(x), N, H, W_, Cc, ptr(y), ptr(amx), Ho, Wo, hip.stream()), "sp_maxpool3s2_fwd_idx")
        ctx.shape = (N, H, W_, Cc, Ho, Wo)
        ctx.save_for_backward(amx)
        return y

    @staticmethod
    def backward(ctx, dy):
        (amx,) = ctx.saved_tensors
        dy = dy.contiguous()
        N, H, W_, Cc, Ho, Wo = ctx.shape
        dx = torch.empty((N, H, W_, Cc), dtype=torch.float32, device=dy.device)
        check(hip.lib().sp_maxpool3s2_bwd_idx(ptr(dy), ptr(amx), N, H, W_, Cc, ptr(dx), Ho, Wo, hip.stream()),
              "sp_maxpool3s2_bwd_idx")
        return dx


def maxpool3s2(x):
    return _MaxPool.apply(x)


# ----------------------------------------------------------------------------------------------------
# decoder
# ----------------------------------------------------------------------------------------------------
class _GateConv(Function):
    """Hg = conv3x3(h, W_h[4C]) with the rank-1 gate terms accumulated into its first 3C columns:
         Hg[b,p,n] += sum_k spcol[b,p,k] * wc[b,n,k]
    (conv3x3(W, spatial (x) semantic) == a 9-tap single-channel conv with the per-sample contracted
    filter W.s; baseline_attention.py:40-50).  h may be None (step 0: h == 0)."""
    @staticmethod
    def forward(ctx, h, w_h, spcol, wc, hw):
        spcol = spcol.contiguous()
        wc = wc.contiguous()
        B, P, KP = spcol.shape
        N3 = wc.shape[1]
        wp = _phys(w_h.detach())
        C4 = wp.shape[0]
        if h is not None:
            h = h.contiguous()
            _, Hm, Wm, Cc = h.shape
            hg = torch.empty((B, Hm, Wm, C4), dtype=torch.float32, device=spcol.device)
            if _b3_pays(B * Hm * Wm, C4, 9 * Cc, Cc):
                hs_ = split_op(h)
                _igemm_b3(hs_, _weight_operand(wp, hs_, None), None, hg, N_img=B, Hi=Hm, Wi=Wm, Kc=Cc, ldx=Cc, Ho=Hm, Wo=Wm, Nout=C4,
                          ldc=C4, ldw=9 * Cc, KH=3, KW=3, pad=1, mode=0)
            else:
                _igemm(h, wp, None, hg, N_img=B, Hi=Hm, Wi=Wm, Kc=Cc, ldx=Cc, Ho=Hm, Wo=Wm, Nout=C4, ldc=C4, ldw=9 * Cc,
                       KH=3, KW=3, pad=1, mode=0)
        else:
            hg = torch.zeros((B, hw[0], hw[1], C4), dtype=torch.float32, device=spcol.device)
        _igemm(spcol, wc, None, hg, N_img=P, Hi=1, Wi=1, Kc=KP, ldx=KP, Ho=1, Wo=1, Nout=N3, ldc=C4, ldw=KP, mode=0, beta=1,
               nbatch=B, sX=P * KP, sW=N3 * KP, sC=P * C4)
        ctx.has_h = h is not None
        ctx.save_for_backward(h, wp, spcol, wc)
        return hg

    @staticmethod
    def backward(ctx, dhg):
        h, wp, spcol, wc = ctx.saved_tensors
        dhg = dhg.contiguous()
        B, P, KP = spcol.shape
        N3 = wc.shape[1]
        C4 = wp.shape[0]
        dh = dw = dsp = dwc = None
        if ctx.has_h:
            _, Hm, Wm, Cc = h.shape
            dys = None
            if ctx.needs_input_grad[0]:
                dh = torch.empty_like(h)
                if _b3_pays(B * Hm * Wm, Cc, 9 * C4, C4):
                    dys = split_op(dhg)
                    _igemm_b3(dys, _weight_operand(wp, dys, None, transposed=True), None, dh, N_img=B, Hi=Hm, Wi=Wm, Kc=C4, ldx=C4, Ho=Hm, Wo=Wm,
                              Nout=Cc, ldc=Cc, ldw=9 * C4, KH=3, KW=3, pad=1, mode=1)
                else:
                    _igemm(dhg, wp, None, dh, N_img=B, Hi=Hm, Wi=Wm, Kc=C4, ldx=C4, Ho=Hm, Wo=Wm, Nout=Cc, ldc=Cc, ldw=Cc,
                           KH=3, KW=3, pad=1, mode=1)
            if ctx.needs_input_grad[1]:
                dwp = torch.empty_like(wp)
                if _w3_pays(B * Hm * Wm, C4, 9 * Cc, Cc):
                    wsch = _wgrad_scheme(Cc, C4)
                    _wgrad_b3(split_op(h, wsch), dys if dys is not None and dys.scheme == wsch else split_op(dhg, wsch), dwp, N_img=B, Hi=Hm, Wi=Wm, Ci=Cc, Ho=Hm,
                              Wo=Wm, Co=C4, ldo=9 * Cc, KH=3, KW=3, pad=1)
                else:
                    _wgrad(h, dhg, dwp, N_img=B, Hi=Hm, Wi=Wm, Ci=Cc, ldx=Cc, Ho=Hm, Wo=Wm, Co=C4, ldy=C4, ldo=9 * Cc,
                           KH=3, KW=3, pad=1)
                dw = dwp.permute(0, 3, 1, 2)
        elif ctx.needs_input_grad[1]:
            dw = torch.zeros_like(wp).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[2]:
            dsp = torch.empty_like(spcol)
            _igemm(dhg, wc, None, dsp, N_img=P, Hi=1, Wi=1, Kc=N3, ldx=C4, Ho=1, Wo=1, Nout=KP, ldc=KP, ldw=KP, mode=1,
                   nbatch=B, sX=P * C4, sW=N3 * KP, sC=P * KP)
        if ctx.needs_input_grad[3]:
            dwc = torch.empty_like(wc)
            _wgrad(spcol, dhg, dwc, N_img=P, Hi=1, Wi=1, Ci=KP, ldx=KP, Ho=1, Wo=1, Co=N3, ldy=C4, ldo=KP, nbatch=B,
                   sX=P * KP, sY=P * C4, sO=N3 * KP)
        return dh, dw, dsp, dwc, None


def gate_conv(h, w_h, spcol, wc, hw):
    return _GateConv.apply(h, w_h, spcol, wc, hw)


class _LstmCell(Function):
    @staticmethod
    def forward(ctx, xg, hg, c_prev):
        xg = xg.contiguous()
        hg = hg.contiguous() if hg is not None else None
        c_prev = c_prev.contiguous() if c_prev is not None else None
        C4 = xg.shape[-1]
        Cc = C4 // 4
        rows = xg.numel() // C4
        shp = xg.shape[:-1] + (Cc,)
        gates = torch.empty_like(xg)
        c = torch.empty(shp, dtype=torch.float32, device=xg.device)
        h = torch.empty(shp, dtype=torch.float32, device=xg.device)
        check(hip.lib().sp_lstm_pointwise_fwd(ptr(xg), ptr(hg), ptr(c_prev), rows, Cc, ptr(gates), ptr(c), ptr(h),
                                              hip.stream()), "sp_lstm_pointwise_fwd")
        ctx.has = (hg is not None, c_prev is not None)
        ctx.save_for_backward(gates, c_prev, c)
        return h, c

    @staticmethod
    def backward(ctx, dh, dc):
        gates, c_prev, c = ctx.saved_tensors
        dh = dh.contiguous() if dh is not None else None
        dc = dc.contiguous() if dc is not None else None
        C4 = gates.shape[-1]
        Cc = C4 // 4
        rows = gates.numel() // C4
        dpre = torch.empty_like(gates)
        dcp = torch.empty_like(c)
        check(hip.lib().sp_lstm_pointwise_bwd(ptr(dh), ptr(dc), ptr(gates), ptr(c_prev), ptr(c), rows, Cc, ptr(dpre),
                                              ptr(dcp), None, hip.stream()), "sp_lstm_pointwise_bwd")
        has_hg, has_c = ctx.has
        return dpre, (dpre if has_hg else None), (dcp if has_c else None)


def lstm_cell(xg, hg, c_prev):
    return _LstmCell.apply(xg, hg, c_prev)


class _LstmCellRank1(Function):
    """ConvLSTM cell with the rank-1 gate terms fused into the pointwise kernel (sp_lstm_rank1_fwd):
         pre[b,p,n] = xg + hg + sum_k spcol[b,p,k] * wc[b,n,k]   (n < 3C: the i/f/o gates)
    xg / hg [B,Hm,Wm,4C] (hg may be None at step 0), spcol [B,P,KP], wc [B,3C,KP]."""
    @staticmethod
    def forward(ctx, xg, hg, c_prev, spcol, wc, step=None, hslot=None):
        # hslot: (DrtBatch, t) -- the hidden state is written into slot t of the batch's [T, B, Hm, Wm, C] buffer (see DrtBatch)
        ctx.step = step
        xg = xg.contiguous()
        hg = hg.contiguous() if hg is not None else None
        c_prev = c_prev.contiguous() if c_prev is not None else None
        spcol = spcol.contiguous()
        wc = wc.contiguous()
        B, P, KP = spcol.shape
        C4 = xg.shape[-1]
        Cc = C4 // 4
        assert xg.numel() == B * P * C4 and wc.shape == (B, 3 * Cc, KP), (xg.shape, spcol.shape, wc.shape)
        shp = xg.shape[:-1] + (Cc,)
        gates = torch.empty_like(xg)
        c = torch.empty(shp, dtype=torch.float32, device=xg.device)
        h = torch.empty(shp, dtype=torch.float32, device=xg.device) if hslot is None else hslot[0].hbuf(hslot[1], c)
        hint = _amax_hint(xg.device)
        check(hip.lib().sp_lstm_rank1_fwd(ptr(xg), ptr(hg), ptr(c_prev), ptr(spcol), ptr(wc), B, P, Cc, KP, ptr(gates), ptr(c),
                                          ptr(h), _hint_ptr(hint), hip.stream()), "sp_lstm_rank1_fwd")
        if hint is not None:
            h._sp_amax = hint
        h._sp_cache = {}                  # h feeds the saliency tap GEMM of this step and the h-gate conv of the next: split once
        ctx.has = (hg is not None, c_prev is not None)
        ctx.fan = getattr(xg, "_sp_fan", None)
        ctx.skip_ok = hg is None and ctx.fan is not None      # dpre's only consumer (xg's fan-in) takes a split operand
        ctx.set_materialize_grads(False)          # the last step's dc stays None instead of a zero tensor the kernel would read
        ctx.cbounds = _cell_bounds(c_prev, c)
        ctx.save_for_backward(gates, c_prev, c, spcol, wc)
        return h, c

    @staticmethod
    def backward(ctx, dh, dc):
        gates, c_prev, c, spcol, wc = ctx.saved_tensors
        dpre, dcp, dsp, dwc = _lstm_rank1_backward(gates, c_prev, c, spcol, wc, dh, dc, ctx.needs_input_grad[3],
                                                   ctx.needs_input_grad[4], ctx.cbounds, skip_fp32=ctx.skip_ok, fan=ctx.fan, step=ctx.step)
        has_hg, has_c = ctx.has
        return dpre, (dpre if has_hg else None), (dcp if has_c else None), dsp, dwc, None, None


def _cell_bounds(c_prev, c):
    """(bound of max|c|, bound of max|c_prev|) from |c_t| <= |c_{t-1}| + 1 (gates in (0,1), |g| < 1); travels on the state tensor"""
    cbp = 0.0 if c_prev is None else getattr(c_prev, "_sp_cbound", None)
    cb = None if cbp is None else cbp + 1.0
    if cb is not None:
        c._sp_cbound = cb
    return cb, cbp




def _split_wcT(wc: torch.Tensor) -> SplitOperand:
    """the rank-1 filters wc [B, 3C, KP] as the 2xfp16 operand [B*KP rows][3C] of their data gradient, one scale per row -- one launch
    (a transposed copy and sp_split2_f16_rows before; same planes, same scales)"""
    B, N3, KP = wc.shape
    wc = wc.contiguous()
    out = torch.empty(2 * wc.numel() + 32, dtype=torch.float16, device=wc.device)
    rscale = torch.empty(B * KP, dtype=torch.float32, device=wc.device)
    check(hip.lib().sp_split2_f16_wT_rows_batched(ptr(wc), B, N3, 1, KP, None, ptr(out), ptr(rscale), hip.stream()),
          "sp_split2_f16_wT_rows_batched")
    return SplitOperand(out, rscale, "f16x2", "rows", None)


def _lstm_rank1_backward(gates, c_prev, c, spcol, wc, dh, dc, need_dsp, need_dwc, cbounds=(None, None), skip_fp32=False, fan=None,
                         step=None):
    """gradients of the cell w.r.t. the gate pre-activations (dpre, carrying its max|.| hint or -- when bounds of max|dh|, max|dc| and
    max|c| are known -- its 2xfp16 split operand, written by the same pass), c_prev, spcol and wc"""
    dh_h = getattr(dh, "_sp_amax", None) if dh is not None else None
    dc_h = getattr(dc, "_sp_amax", None) if dc is not None else None
    dh = dh.contiguous() if dh is not None else None
    dc = dc.contiguous() if dc is not None else None
    B, P, KP = spcol.shape
    C4 = gates.shape[-1]
    Cc = C4 // 4
    N3 = 3 * Cc
    rows = gates.numel() // C4
    dpre = torch.empty_like(gates)
    dcp = torch.empty_like(c)
    hint, chint = _amax_hint(gates.device), _amax_hint(gates.device)
    rank1_split_ok = P % 256 == 0 and N3 % 32 == 0 and KP <= 32 and KP % 4 == 0 and not THROUGHPUT_MODE
    rc = rows_ctx(step, B)          # samples without loss gradient at this decode step: zero outputs, inputs not read
    emit = (LSTM_BWD_SPLIT and hint is not None and Cc % 256 == 0 and cbounds[0] is not None and (dh is None or dh_h is not None)
            and (dc is None or dc_h is not None) and _scheme_for(C4) == "f16x2")
    if emit:
        FUSION_COUNTS["lstm_bwd_split"] += 1
        planes = torch.empty(2 * dpre.numel() + 32, dtype=torch.float16, device=dpre.device)
        # skip_fp32 (the caller's other consumers of dpre read its split form): when the two rank-1 gradients do too, the fp32
        # tensor is allocated (autograd wants one) but never written -- 671 MB per step at the benchmark size
        skip = (skip_fp32 and fan is not None and LSTM_SKIP_DPRE and (not need_dsp or (RANK1_DSP_SPLIT and rank1_split_ok))
                and (not need_dwc or (RANK1_DWC_SPLIT and rank1_split_ok)))
        if skip:
            FUSION_COUNTS["lstm_skip_dpre"] += 1
            dpre._sp_skipped = True          # guards every fp32 reader this process owns (_fp32_required); the fan-in reads the record
        check(hip.lib().sp_lstm_pointwise_bwd_rows(ptr(dh), ptr(dc), ptr(gates), ptr(c_prev), ptr(c), rows, Cc, None if skip else ptr(dpre), ptr(dcp),
                                                   None, _hint_ptr(chint), _hint_ptr(dh_h), _hint_ptr(dc_h), float(cbounds[0]),
                                                   float(cbounds[1]), ptr(planes), ptr(hint), ptr(rc.last) if rc is not None else None,
                                                   int(step) if rc is not None else 0, P, hip.stream()),
              "sp_lstm_pointwise_bwd_rows")
        dpre._sp_amax = hint
        dpre._sp_cache = {"f16x2": SplitOperand(planes, hint, "f16x2")}
        if rc is not None:
            dpre._sp_cache["f16x2"].rows = (rc, int(step))
        if skip:
            fan[0][fan[1]] = dpre._sp_cache["f16x2"]      # xg's fan-in takes this contribution from the record, by alias index
    else:
        check(hip.lib().sp_lstm_pointwise_bwd_split(ptr(dh), ptr(dc), ptr(gates), ptr(c_prev), ptr(c), rows, Cc, ptr(dpre), ptr(dcp),
                                                    _hint_ptr(hint), _hint_ptr(chint), None, None, 0.0, 0.0, None, None,
                                                    hip.stream()), "sp_lstm_pointwise_bwd_split")
        if hint is not None:
            dpre._sp_amax = hint
    if chint is not None:
        dcp._sp_amax = chint               # max|dc_prev|: the previous step's cell backward bounds its operand with it
    dsp = dwc = None
    L = hip.lib()
    if (need_dsp and need_dwc and emit and RANK1_FUSED and RANK1_DSP_SPLIT and RANK1_DWC_SPLIT and rank1_split_ok
            and L.sp_rank1_grads_applies(B, P, N3, KP, C4)):
        # both gradients of the rank-1 gate term from ONE pass over the split dpre the cell backward just wrote (csrc/rank1_grads.hip): dsp on
        # the matrix pipe against the split form of wc^T, dwc on the vector pipe from the same LDS tiles, spcol as it is -- instead of two
        # launches of the big GEMM kernels that each re-read the 503 MB of planes, and six small launches preparing their operands
        FUSION_COUNTS["rank1_fused"] += 1
        ys = dpre._sp_cache["f16x2"]
        ws = _split_wcT(wc)             # [B][KP][3C]: K contiguous, one scale per row
        dsp, dwc = torch.empty_like(spcol), torch.empty_like(wc)
        wsp = hip.workspace(L.sp_rank1_grads_workspace(B, P, N3, KP), dpre.device, slot=0)
        check(L.sp_rank1_grads_f16x2(ptr(ys.buf), ptr(ys.scale), C4, ptr(ws.buf), ptr(ws.scale), ptr(spcol), B, P, N3, KP, ptr(dsp), ptr(dwc),
                                     ptr(wsp), ptr(rc.last) if rc is not None else None, int(step) if rc is not None else 0, hip.stream()),
              "sp_rank1_grads_f16x2")
        return dpre, dcp, dsp, dwc
    if need_dsp:
        dsp = torch.empty_like(spcol)
        if emit and RANK1_DSP_SPLIT and rank1_split_ok:
            # gradient of the spatial-memory taps, dsp[b] = dpre[b][:, :3C] x wc[b]: a batched pointwise GEMM with one weight set
            # per sample, on the split dpre the cell backward just wrote (its first 3C of 4C channels per row) -- the fp32-MFMA
            # batched GEMM it replaces read the fp32 dpre (0.5 GB) at ~2 TB/s
            FUSION_COUNTS["rank1_dsp_split"] += 1
            xs = dpre._sp_cache["f16x2"]
            ws = _split_wcT(wc)             # [B][KP][3C]: K contiguous, one scale per row
            d = ConvDesc(P, 1, 1, N3, C4, 1, 1, KP, KP, 1, 1, 1, 0, 1, 0, N3, 1.0, 0, 0, B, P * C4, KP * N3, P * KP, 0, None)
            d.w_scale_rows = 1
            if rc is not None:          # one item per sample: items without loss gradient at this step are zero tiles
                d.row_last, d.row_step = rc.last.data_ptr(), int(step)
            check(hip.lib().sp_conv_igemm_f16x2(C.byref(d), ptr(xs.buf), ptr(xs.scale), ptr(ws.buf), ptr(ws.scale), None, ptr(dsp),
                                                hip.stream()), "sp_conv_igemm_f16x2 (batched)")
        else:
            _igemm(dpre, wc, None, dsp, N_img=P, Hi=1, Wi=1, Kc=N3, ldx=C4, Ho=1, Wo=1, Nout=KP, ldc=KP, ldw=KP, mode=1,
                   nbatch=B, sX=P * C4, sW=N3 * KP, sC=P * KP)
    if need_dwc:
        dwc = torch.empty_like(wc)
        if emit and RANK1_DWC_SPLIT and rank1_split_ok:
            # filter gradient of the rank-1 gate term, dwc[b] = dpre[b][:, :3C]^T x spcol[b]: one TN GEMM per sample (K = the sample's
            # pixels) on the split dpre; spcol's KP tap columns are padded to 32 for the 16-channel groups of the split operand, the
            # kernel stores the first KP columns.  The VALU kernel it replaces read the fp32 dpre at ~2 TB/s (250 us per launch)
            FUSION_COUNTS["rank1_dwc_split"] += 1
            ys = dpre._sp_cache["f16x2"]
            xs = split_op(torch.nn.functional.pad(spcol, (0, 32 - KP)), "f16x2")
            d = hip.WgradDesc(1, P // 64, 64, 32, 32, P // 64, 64, N3, C4, 1, 1, 1, 0, 1, KP, 0, 1.0, B, P * 32, P * C4, N3 * KP)
            if rc is not None:
                d.row_last, d.row_step = rc.last.data_ptr(), int(step)
            check(L.sp_conv_wgrad_f16x2(C.byref(d), ptr(xs.buf), ptr(xs.scale), ptr(ys.buf), ptr(ys.scale), ptr(dwc), None, hip.stream()),
                  "sp_conv_wgrad_f16x2 (batched)")
        elif KP <= 24:
            ws = hip.workspace(L.sp_rank1_dwc_workspace(B, P, N3, KP), dpre.device, slot=0)
            check(L.sp_rank1_dwc(ptr(dpre), ptr(spcol), B, P, C4, N3, KP, ptr(ws), ptr(dwc), hip.stream()), "sp_rank1_dwc")
        else:
            _wgrad(spcol, dpre, dwc, N_img=P, Hi=1, Wi=1, Ci=KP, ldx=KP, Ho=1, Wo=1, Co=N3, ldy=C4, ldo=KP, nbatch=B,
                   sX=P * KP, sY=P * C4, sO=N3 * KP)
    return dpre, dcp, dsp, dwc


def lstm_cell_rank1(xg, hg, c_prev, spcol, wc, step=None, hslot=None):
    return _LstmCellRank1.apply(xg, hg, c_prev, spcol, wc, step, hslot)




def gateconv_lstm_fusable(h, w_h, spcol) -> bool:
    """the fused h-gate conv + cell kernel (sp_gateconv_lstm_f16x2) applies: 2xfp16 back-end, 3x3 stride-1 gate conv on
    C % 32 == 0 channels, P % 256 == 0 pixels per sample (a 256-pixel tile inside one sample), KP <= 32"""
    if not (FUSE_GATE_LSTM and USE_BF16X3 and not THROUGHPUT_MODE and h is not None):
        return False
    N, H, W_, Ci = h.shape
    B, P, KP = spcol.shape
    Co, _, KH, KW = w_h.shape
    return (_scheme_for(Ci) == "f16x2" and Ci % 32 == 0 and Co == 4 * Ci and (KH, KW) == (3, 3) and P == H * W_
            and P % 256 == 0 and KP <= 32 and _b3_pays(N * P, Co, 9 * Ci, Ci, a_elems=h.numel(), free_a=True))


class _GateConvLstm(Function):
    """One ConvLSTM step t >= 1 in ONE kernel: the cell (with the rank-1 gate term) is the epilogue of the h-gate conv
    (sp_gateconv_lstm_f16x2), so the [B,Hm,Wm,4C] h-gate tensor is never materialised.  Backward = _LstmCellRank1's followed by
    _Conv2d's (same kernels as the unfused pair)."""
    @staticmethod
    def forward(ctx, h_prev, w_h, xg, c_prev, spcol, wc, wcache, step=None, hslot=None):
        ctx.step = step
        h_prev, xg, c_prev = h_prev.contiguous(), xg.contiguous(), c_prev.contiguous()
        spcol, wc = spcol.contiguous(), wc.contiguous()
        N, H, W_, Ci = h_prev.shape
        B, P, KP = spcol.shape
        wp = _phys(w_h.detach())
        Co, KH, KW, _ = wp.shape
        assert xg.numel() == B * P * Co and wc.shape == (B, 3 * Ci, KP) and N == B, (xg.shape, spcol.shape, wc.shape)
        xs = split_op(h_prev)
        wsplit = _weight_operand(wp, xs, wcache)
        if xs.kind != "scalar":
            raise RuntimeError("scanpaths_amd: the fused gate conv expects the hidden state's operand with a per-tensor scale")
        gates = torch.empty_like(xg)
        c = torch.empty_like(c_prev)
        h = torch.empty_like(c_prev) if hslot is None else hslot[0].hbuf(hslot[1], c_prev)          # (DrtBatch, t): see _LstmCellRank1
        hint = _amax_hint(xg.device)
        d = ConvDesc(N, H, W_, Ci, Ci, H, W_, Co, Co, KH, KW, 1, 1, 1, 0, KH * KW * Ci, 1.0, 0, 0, 1, 0, 0, 0, 0, None)
        d.w_scale_rows = int(wsplit.kind == "rows")
        cbounds = _cell_bounds(c_prev, c)
        # |h| = |o * c| <= |c| <= t + 1: with that bound the epilogue writes h's split operand itself (no max|h| pass, no split pass)
        hplanes = torch.empty(2 * h.numel() + 32, dtype=torch.float16, device=h.device) \
            if (LSTM_H_PLANES and hint is not None and cbounds[0] is not None and Ci % 16 == 0) else None

        def launch():
            check(hip.lib().sp_gateconv_lstm_f16x2(C.byref(d), ptr(xs.buf), ptr(xs.scale), ptr(wsplit.buf), ptr(wsplit.scale),
                                                   ptr(xg), ptr(c_prev), ptr(spcol), ptr(wc), P, KP, ptr(gates), ptr(c), ptr(h),
                                                   None if hplanes is not None else _hint_ptr(hint), ptr(hplanes),
                                                   ptr(hint) if hplanes is not None else None,
                                                   float(cbounds[0]) if hplanes is not None else 0.0, hip.stream()),
                  "sp_gateconv_lstm_f16x2")
        if hip.TIMER is None:
            launch()
        else:
            hip.TIMER.bracket(("h2_fwd", N * P, Co, KH * KW * Ci, f"{KH}x{KW}", 1), 2.0 * N * P * Co * KH * KW * Ci, launch)
        FUSION_COUNTS["gateconv_lstm"] += 1
        FUSION_COUNTS["gateconv_lstm_hplanes"] += int(hplanes is not None)
        if hint is not None:
            h._sp_amax = hint
        h._sp_cache = {"f16x2": SplitOperand(hplanes, hint, "f16x2")} if hplanes is not None else {}
        ctx.set_materialize_grads(False)
        ctx.cbounds = cbounds
        defer = wcache.get("defer") if isinstance(wcache, dict) else None
        ctx.defer_final = defer.claim() if (defer is not None and ctx.needs_input_grad[1]) else False
        keep = ctx.needs_input_grad[1] and _w3_pays(N * P, Co, KH * KW * Ci, Ci, free_splits=True) \
            and xs.scheme == _wgrad_scheme(Ci, Co)
        ctx.xs_scheme = (xs.scheme, xs.kind) if keep else None
        ctx.wcache = wcache
        # the gate gradient dpre of this step has three consumers: xg's fan-in, the h-gate conv's data and weight gradient.  When all
        # of them read its split form, its fp32 form is never written (_lstm_rank1_backward skip_fp32)
        ctx.fan = getattr(xg, "_sp_fan", None)
        ctx.h_events = getattr(h_prev, "_sp_fan_ev", None)      # (events dict of h_prev's fan-out, this consumer's index) or None
        ctx.skip_ok = (ctx.fan is not None and _scheme_for(Co) == "f16x2"
                       and (not ctx.needs_input_grad[0] or _b3_pays(N * P, Ci, KH * KW * Co, Co, a_elems=xg.numel(), free_a=True))
                       and (not ctx.needs_input_grad[1] or (keep and _wgrad_scheme(Ci, Co) == "f16x2")))
        ctx.save_for_backward(gates, c_prev, c, spcol, wc, h_prev, wp, xs.buf if keep else None, xs.scale if keep else None)
        return h, c

    @staticmethod
    def backward(ctx, dh, dc):
        gates, c_prev, c, spcol, wc, h_prev, wp, xs_buf, xs_scale = ctx.saved_tensors
        dpre, dcp, dsp, dwc = _lstm_rank1_backward(gates, c_prev, c, spcol, wc, dh, dc, ctx.needs_input_grad[4],
                                                   ctx.needs_input_grad[5], ctx.cbounds, skip_fp32=ctx.skip_ok, fan=ctx.fan, step=ctx.step)
        xs = SplitOperand(xs_buf, xs_scale, *ctx.xs_scheme) if xs_buf is not None else None
        if (ASYNC_DGRAD and ctx.h_events is not None and ctx.needs_input_grad[0] and dpre.is_cuda
                and not torch.cuda.is_current_stream_capturing()):
            # The data gradient of the h-gate conv (1.5 TFLOP, 1.3-3.3 ms) has ONE reader, the fan-in of h_{t-1}'s three gradients; the
            # other results of this node (gradients of the spatial taps and of the contracted filters) feed ~25 small launches -- the
            # backward of the memory update and of step t - 1's heads, 1.2 ms that cannot fill the chip -- whose results meet the data
            # gradient only in that same fan-in.  So the GEMM goes to the side stream (it waits for everything enqueued so far: dpre's
            # planes) and the current stream carries on with the small launches; the fan-in waits for the event left in its `events`.
            # The weight gradient stays on the current stream: recorded (deferred to one launch over all applications, issued by the
            # application that runs last in backward) or computed -- it reads what the forward and the cell backwards wrote.
            main, side = torch.cuda.current_stream(), hip.side_stream(dpre.device)
            # process-global cached objects the side call may draw (single-use scale slots of the hint pool, the constant 1.0) are
            # created HERE, under the current stream: a pool zero-filled under the side stream would race the current stream's atomicMax
            # into its slots
            _reserve_hints(dpre.device, 8)
            _one(dpre.device)
            ready = torch.cuda.Event()
            ready.record(main)
            cached_before = set(ctx.wcache) if isinstance(ctx.wcache, dict) else set()
            with torch.cuda.stream(side):
                side.wait_event(ready)
                dhp, _ = _conv_backward(h_prev, wp, dpre, xs, 1, 1, 1, ctx.wcache, True, False, step=ctx.step)
                done = torch.cuda.Event()
                done.record(side)
            dhp.record_stream(main)                      # allocated under the side stream, read (and freed) on the current one
            # ... and the other way round: what the GEMM READS was allocated under the current stream and is freed there -- by autograd, as
            # soon as this node returns, unless another consumer happens to hold it (xg's fan-in, the deferred weight gradient).  The
            # caching allocator must not hand those blocks to a current-stream kernel while the 1.3-3.3 ms GEMM is still reading them.
            rc_ = rows_ctx(ctx.step, h_prev.shape[0])
            planes = (getattr(dpre, "_sp_cache", None) or {}).get("f16x2")
            for t_ in (dpre, h_prev, wp, planes.buf if planes is not None else None, planes.scale if planes is not None else None,
                       rc_.last if rc_ is not None else None):
                if t_ is not None and t_.is_cuda:
                    t_.record_stream(side)
            if isinstance(ctx.wcache, dict):             # a weight operand first built under the side stream may later meet the current one
                for k_ in set(ctx.wcache) - cached_before:
                    op_ = ctx.wcache[k_]
                    if isinstance(op_, SplitOperand):
                        for t_ in (op_.buf, op_.scale):
                            if t_ is not None:
                                t_.record_stream(main)
            events, idx = ctx.h_events
            events[idx] = done
            FUSION_COUNTS["async_dgrad"] += 1
            dw = None
            if ctx.needs_input_grad[1]:
                _, dw = _conv_backward(h_prev, wp, dpre, xs, 1, 1, 1, ctx.wcache, False, True, defer_final=ctx.defer_final, step=ctx.step)
            return dhp, dw, dpre, dcp, dsp, dwc, None, None, None
        dhp, dw = _conv_backward(h_prev, wp, dpre, xs, 1, 1, 1, ctx.wcache, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                 defer_final=ctx.defer_final, step=ctx.step)
        return dhp, dw, dpre, dcp, dsp, dwc, None, None, None


def gateconv_lstm(h_prev, w_h, xg, c_prev, spcol, wc, wcache=None, step=None, hslot=None):
    return _GateConvLstm.apply(h_prev, w_h, xg, c_prev, spcol, wc, wcache, step, hslot)


class _Im2col(Function):
    """maps [S, R, H, W] -> col [R, H*W, KP]; stream s occupies columns [9s, 9s+9), the rest is zero."""
    @staticmethod
    def forward(ctx, maps, KP):
        maps = maps.contiguous()
        S, R, H, W_ = maps.shape
        col = torch.empty((R, H * W_, KP), dtype=torch.float32, device=maps.device)
        check(hip.lib().sp_im2col3x3_multi(ptr(maps), S, R, H, W_, KP, ptr(col), hip.stream()), "sp_im2col3x3_multi")      # one launch, zero padding columns included
        ctx.shape = (S, R, H, W_, KP)
        return col

    @staticmethod
    def backward(ctx, dcol):
        S, R, H, W_, KP = ctx.shape
        dcol = dcol.contiguous()
        dm = torch.empty((S, R, H, W_), dtype=torch.float32, device=dcol.device)
        check(hip.lib().sp_col2im3x3_multi(ptr(dcol), S, R, H, W_, KP, ptr(dm), hip.stream()), "sp_col2im3x3_multi")
        return dm, None


def im2col3x3(maps, KP):
    return _Im2col.apply(maps, KP)


class _ListAtt(Function):
    """L [T,R,D], u [D] -> mem [R,D] = sum_t softmax_t(<L[t,r],u>) L[t,r]"""
    @staticmethod
    def forward(ctx, Lst, u):
        Lst = Lst.contiguous()
        u = u.contiguous()
        T, R, D = Lst.shape
        mem = torch.empty((R, D), dtype=torch.float32, device=Lst.device)
        alpha = torch.empty((T, R), dtype=torch.float32, device=Lst.device)
        check(hip.lib().sp_listatt_fwd(ptr(Lst), ptr(u), T, R, D, ptr(mem), ptr(alpha), hip.stream()), "sp_listatt_fwd")
        ctx.save_for_backward(Lst, u, alpha)
        return mem

    @staticmethod
    def backward(ctx, dmem):
        Lst, u, alpha = ctx.saved_tensors
        dmem = dmem.contiguous()
        T, R, D = Lst.shape
        dL = torch.empty_like(Lst)
        dup = torch.empty((R, D), dtype=torch.float32, device=Lst.device)
        check(hip.lib().sp_listatt_bwd(ptr(dmem), ptr(Lst), ptr(u), ptr(alpha), T, R, D, ptr(dL), ptr(dup), hip.stream()),
              "sp_listatt_bwd")
        du = _colsum_any(dup, D) if ctx.needs_input_grad[1] else None
        return dL, du


def list_attention(Lst, u):
    return _ListAtt.apply(Lst, u)


class _MulRelu(Function):
    """a [S, n] (any trailing shape), b [n]  ->  relu(a * b)"""
    @staticmethod
    def forward(ctx, a, b):
        a = a.contiguous()
        b = b.contiguous()
        out = torch.empty_like(a)
        check(hip.lib().sp_mulrelu_fwd(ptr(a), ptr(b), a.numel(), b.numel(), ptr(out), hip.stream()), "sp_mulrelu_fwd")
        ctx.save_for_backward(a, b, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        a, b, out = ctx.saved_tensors
        dout = dout.contiguous()
        da = torch.empty_like(a)
        dbp = torch.empty_like(a)
        check(hip.lib().sp_mulrelu_bwd(ptr(dout), ptr(a), ptr(b), ptr(out), a.numel(), b.numel(), ptr(da), ptr(dbp),
                                       hip.stream()), "sp_mulrelu_bwd")
        S = a.numel() // b.numel()
        parts = dbp.view(S, -1)
        db = parts[0]
        for s in range(1, S):
            db = _add_raw(db, parts[s])
        return da, db.view(b.shape)


def mul_relu(a, b):
    return _MulRelu.apply(a, b)


class _SemPool(Function):
    """amaps [S,B,P], vf [B,P,C] -> relu(mean_p(amaps * vf)) [B,S,C]   (get_channel_semantic + ReLU); sbc: the result as [S,B,C], the
    row order of the embedding behind it (no transposed copy in forward, none of its gradient in backward: 2 launches per decode step)"""
    @staticmethod
    def forward(ctx, amaps, vf, step=None, sbc=False):
        amaps, vf = amaps.contiguous(), vf.contiguous()
        S, B, P = amaps.shape
        Cc = vf.shape[-1]
        L = hip.lib()
        out = torch.empty((S, B, Cc) if sbc else (B, S, Cc), dtype=torch.float32, device=vf.device)
        ws = hip.workspace(L.sp_sempool_workspace(S, B, P, Cc), vf.device, slot=0)
        check(L.sp_sempool_fwd_sbc(ptr(amaps), ptr(vf), S, B, P, Cc, 1.0 / P, ptr(ws), ptr(out), int(sbc), hip.stream()), "sp_sempool_fwd_sbc")
        ctx.save_for_backward(amaps, vf, out)
        ctx.step, ctx.sbc = step, sbc
        return out

    @staticmethod
    def backward(ctx, dout):
        amaps, vf, out = ctx.saved_tensors
        S, B, P = amaps.shape
        Cc = vf.shape[-1]
        da, dvf = torch.empty_like(amaps), torch.empty_like(vf)
        rc = rows_ctx(ctx.step, B)      # memory update `step` feeds decode steps >= step only: samples whose last loss step is earlier get zeros
        check(hip.lib().sp_sempool_bwd_rows_sbc(ptr(dout.contiguous()), ptr(out), ptr(amaps), ptr(vf), S, B, P, Cc, 1.0 / P, ptr(da), ptr(dvf),
                                                ptr(rc.last) if rc is not None else None, int(ctx.step) if rc is not None else 0, int(ctx.sbc),
                                                hip.stream()), "sp_sempool_bwd_rows_sbc")
        if rc is not None:
            dvf._sp_rows = (rc, int(ctx.step))      # hint for vf's gradient fan-in (dead samples' rows are exact zeros): F._FanOut
        return da, dvf, None, None


def semantic_pool(amaps, vf, step=None, sbc=False):
    return _SemPool.apply(amaps, vf, step, sbc)


class _RowMean(Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        Cc = x.shape[-1]
        M = x.numel() // Cc
        out = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
        check(hip.lib().sp_rowsum(ptr(x), M, Cc, 1.0 / Cc, ptr(out), hip.stream()), "sp_rowsum")
        ctx.shape = x.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        Cc = ctx.shape[-1]
        dx = torch.zeros(ctx.shape, dtype=torch.float32, device=dout.device)
        check(hip.lib().sp_rowsum_bwd(ptr(dout), dx.numel() // Cc, Cc, 1.0 / Cc, ptr(dx), hip.stream()), "sp_rowsum_bwd")
        return dx


def channel_mean(x):
    return _RowMean.apply(x)


class _SelectRows(Function):
    @staticmethod
    def forward(ctx, a, b, sel):
        a = a.contiguous()
        b = b.contiguous()
        rows = a.shape[0]
        ln = a.numel() // rows
        out = torch.empty_like(a)
        sel8 = sel.to(torch.uint8).contiguous()
        check(hip.lib().sp_select_rows(ptr(a), ptr(b), ptr(sel8), rows, ln, ptr(out), hip.stream()), "sp_select_rows")
        ctx.save_for_backward(sel8)
        return out

    @staticmethod
    def backward(ctx, dout):
        (sel8,) = ctx.saved_tensors
        dout = dout.contiguous()
        rows = dout.shape[0]
        ln = dout.numel() // rows
        da = torch.empty_like(dout)
        db = torch.empty_like(dout)
        check(hip.lib().sp_select_rows_bwd(ptr(dout), ptr(sel8), rows, ln, ptr(da), ptr(db), hip.stream()),
              "sp_select_rows_bwd")
        return da, db, None


def select_rows(a, b, sel):
    return _SelectRows.apply(a, b, sel)


class _HeadFinish(Function):
    """Z [B,Hm,Wm,nh*zc] -> logits [nh,B,1+P], amap [nh,B,P], mu [nh,B], sigma2 [nh,B].
    dpre None: zc = HC and Z carries the 49 duration-tap columns; dpre [nh,B,S]: Z carries the 2 map columns per head only."""
    @staticmethod
    def forward(ctx, Z, cb, w2, b2, nheads, HC, softmax, per_sample, dpre):
        Z = Z.contiguous()
        zc = HC if dpre is None else 2
        if dpre is not None:
            dpre = dpre.contiguous()
        cb = cb.contiguous()
        w2c = w2.detach().contiguous()
        b2 = b2.contiguous()
        B, Hm, Wm, ldz = Z.shape
        P = Hm * Wm
        dev = Z.device
        dh, dw = (Hm + 4 - 7) // 5 + 1, (Wm + 4 - 7) // 5 + 1
        logits = torch.empty((nheads, B, P + 1), dtype=torch.float32, device=dev)
        amap = torch.empty((nheads, B, P), dtype=torch.float32, device=dev)
        mu = torch.empty((nheads, B), dtype=torch.float32, device=dev)
        s2 = torch.empty((nheads, B), dtype=torch.float32, device=dev)
        drt = torch.empty((nheads, B, dh * dw), dtype=torch.float32, device=dev)
        check(hip.lib().sp_head_finish_fwd(ptr(Z), B, Hm, Wm, ldz, nheads, HC, ptr(cb), int(per_sample), ptr(w2c), ptr(b2),
                                           int(softmax),
                                           ptr(logits), ptr(amap), ptr(mu), ptr(s2), ptr(drt), ptr(dpre), zc, hip.stream()),
              "sp_head_finish_fwd")
        ctx.cfg = (B, Hm, Wm, ldz, nheads, HC, softmax, dh, dw, tuple(w2.shape), per_sample, tuple(cb.shape), zc)
        ctx.save_for_backward(logits, amap, s2, drt, w2c)
        return logits, amap, mu, s2

    @staticmethod
    def backward(ctx, dlogits, damap, dmu, ds2):
        B, Hm, Wm, ldz, nheads, HC, softmax, dh, dw, w2shape, per_sample, cbshape, zc = ctx.cfg
        logits, amap, s2, drt, w2c = ctx.saved_tensors
        dev = logits.device
        zf = lambda t, ref: (t.contiguous() if t is not None else torch.zeros_like(ref))
        dlogits = zf(dlogits, logits)
        dmu = zf(dmu, s2)
        ds2 = zf(ds2, s2)
        damap = damap.contiguous() if damap is not None else None
        S = dh * dw
        dZ = torch.empty((B, Hm, Wm, ldz), dtype=torch.float32, device=dev)
        if ldz != nheads * zc:
            dZ.zero_()
        ddpre = torch.empty((nheads, B, S), dtype=torch.float32, device=dev) if zc != HC else None
        dcbp = torch.empty((B, nheads * HC), dtype=torch.float32, device=dev)
        dw2p = torch.empty((B, nheads * 2 * S), dtype=torch.float32, device=dev)
        db2p = torch.empty((B, nheads * 2), dtype=torch.float32, device=dev)
        check(hip.lib().sp_head_finish_bwd(ptr(dlogits), ptr(damap), ptr(dmu), ptr(ds2), ptr(logits), ptr(amap), ptr(s2),
                                           ptr(drt), B, Hm, Wm, ldz, nheads, HC, ptr(w2c), int(softmax), ptr(dZ), ptr(dcbp),
                                           ptr(dw2p), ptr(db2p), ptr(ddpre), zc, hip.stream()), "sp_head_finish_bwd")
        dcb = dcbp.view(cbshape) if per_sample else _colsum_any(dcbp, nheads * HC).view(nheads, HC)
        dw2 = _colsum_any(dw2p, nheads * 2 * S).view(nheads, 2 * S)
        db2 = _colsum_any(db2p, nheads * 2).view(nheads, 2)
        dw2s, db2s = dw2[0], db2[0]
        for k in range(1, nheads):          # drt_layer_2 is shared by the heads
            dw2s = _add_raw(dw2s, dw2[k])
            db2s = _add_raw(db2s, db2[k])
        return dZ, dcb, dw2s.reshape(w2shape), db2s.reshape(2), None, None, None, None, ddpre


def head_finish(Z, cb, w2, b2, nheads, HC, softmax, per_sample=False, dpre=None):
    return _HeadFinish.apply(Z, cb, w2, b2, nheads, HC, softmax, per_sample, dpre)


class _HeadSal(Function):
    """The saliency part of predict_head alone (sp_head_finish_parts_*, parts = 1): Z2 [B,Hm,Wm,nh*2] -> logits [nh,B,1+P], amap [nh,B,P].
    The decode loop evaluates it per step -- the action map feeds the next memory update -- while the duration part (mu, sigma2), which
    nothing in the recurrence reads, is evaluated once for all T steps behind the loop (drt_heads_batched)."""
    @staticmethod
    def forward(ctx, Z, cb, nheads, HC, softmax, per_sample):
        Z = Z.contiguous()
        cb = cb.contiguous()
        B, Hm, Wm, ldz = Z.shape
        P = Hm * Wm
        logits = torch.empty((nheads, B, P + 1), dtype=torch.float32, device=Z.device)
        amap = torch.empty((nheads, B, P), dtype=torch.float32, device=Z.device)
        check(hip.lib().sp_head_finish_parts_fwd(ptr(Z), B, Hm, Wm, ldz, nheads, HC, ptr(cb), int(per_sample), None, None, int(softmax),
                                                 ptr(logits), ptr(amap), None, None, None, None, 2, 1, hip.stream()), "sp_head_finish_parts_fwd")
        ctx.cfg = (B, Hm, Wm, ldz, nheads, HC, softmax, per_sample, tuple(cb.shape))
        ctx.save_for_backward(logits, amap)
        return logits, amap

    @staticmethod
    def backward(ctx, dlogits, damap):
        B, Hm, Wm, ldz, nheads, HC, softmax, per_sample, cbshape = ctx.cfg
        logits, amap = ctx.saved_tensors
        if dlogits is None:
            dlogits = torch.zeros_like(logits)
        # the gradient of step t's logits is a slice of the stacked outputs' gradient [nh, B, T, 1+P]: rows (head, sample) a constant
        # distance apart -- read in place (a contiguous copy per decode step before)
        if not (dlogits.stride(2) == 1 and dlogits.stride(0) == B * dlogits.stride(1) and dlogits.stride(1) >= dlogits.shape[2]):
            dlogits = dlogits.contiguous()
        damap = damap.contiguous() if damap is not None else None
        dZ = torch.empty((B, Hm, Wm, ldz), dtype=torch.float32, device=logits.device)
        if ldz != nheads * 2:
            dZ.zero_()
        dcbp = torch.empty((B, nheads * HC), dtype=torch.float32, device=logits.device)
        check(hip.lib().sp_head_finish_parts_bwd_ld(ptr(dlogits), dlogits.stride(1), ptr(damap), None, None, ptr(logits), ptr(amap), None, None,
                                                    B, Hm, Wm, ldz, nheads, HC, None, int(softmax), ptr(dZ), ptr(dcbp), None, None, None, 2, 1,
                                                    None, hip.stream()), "sp_head_finish_parts_bwd_ld")
        dcb = dcbp.view(cbshape) if per_sample else _colsum_any(dcbp, nheads * HC).view(nheads, HC)
        return dZ, dcb, None, None, None, None


def head_sal(Z, cb, nheads, HC, softmax, per_sample=False):
    return _HeadSal.apply(Z, cb, nheads, HC, softmax, per_sample)


# ---- predict_head without the dense 5x5 GEMM (csrc/head_direct.hip) ---------------------------------------------
def head_num_classes(Hm: int, Wm: int) -> int:
    n = hip.lib().sp_head_num_classes(Hm, Wm)
    if n <= 0:
        raise ValueError(f"unsupported map size {Hm}x{Wm} for the duration branch")
    return n


class _Compose11(Function):
    """G [nh*HC,C,5,5] (physical [nh*HC,5,5,C]), cb [nh,HC] -> W11 [nh,ncls,121,C], cbsum [nh,ncls]"""
    @staticmethod
    def forward(ctx, G, cb, nheads, HC, hw):
        Gp = _phys(G.detach())
        cb = cb.contiguous()
        C_ = Gp.shape[3]
        Hm, Wm = hw
        ncls = head_num_classes(Hm, Wm)
        W11 = torch.empty((nheads, ncls, 121, C_), dtype=torch.float32, device=Gp.device)
        cbsum = torch.empty((nheads, ncls), dtype=torch.float32, device=Gp.device)
        check(hip.lib().sp_head_compose11_fwd(ptr(Gp), ptr(cb), nheads, HC, C_, Hm, Wm, ptr(W11), ptr(cbsum), hip.stream()),
              "sp_head_compose11_fwd")
        ctx.cfg = (nheads, HC, C_, Hm, Wm)
        return W11, cbsum

    @staticmethod
    def backward(ctx, dW11, dcbsum):
        nheads, HC, C_, Hm, Wm = ctx.cfg
        dW11 = dW11.contiguous()
        dcbsum = dcbsum.contiguous() if dcbsum is not None else torch.zeros(dW11.shape[:2], device=dW11.device)
        dGp = torch.empty((nheads * HC, 5, 5, C_), dtype=torch.float32, device=dW11.device)
        dcb = torch.empty((nheads, HC), dtype=torch.float32, device=dW11.device)
        check(hip.lib().sp_head_compose11_bwd(ptr(dW11), ptr(dcbsum), nheads, HC, C_, Hm, Wm, ptr(dGp), ptr(dcb), hip.stream()),
              "sp_head_compose11_bwd")
        return dGp.permute(0, 3, 1, 2), dcb, None, None, None


def compose11(G, cb, nheads, HC, hw):
    return _Compose11.apply(G, cb, nheads, HC, hw)


class _SalGather(Function):
    """tap partials T [B,Hm,Wm,ldt] -> maps Z2 [B,Hm,Wm,2*nsel]; hmap int32 [B,nsel] = source head of each output slot"""
    @staticmethod
    def forward(ctx, T, hmap, nsel, nsrc, step=None):
        T = T.contiguous()
        B, Hm, Wm, ldt = T.shape
        Z2 = torch.empty((B, Hm, Wm, 2 * nsel), dtype=torch.float32, device=T.device)
        check(hip.lib().sp_sal_gather_fwd(ptr(T), B, Hm, Wm, ldt, nsel, ptr(hmap), ptr(Z2), hip.stream()), "sp_sal_gather_fwd")
        ctx.cfg = (B, Hm, Wm, ldt, nsel, nsrc)
        ctx.step = step
        ctx.save_for_backward(hmap)
        return Z2

    @staticmethod
    def backward(ctx, dZ2):
        B, Hm, Wm, ldt, nsel, nsrc = ctx.cfg
        hmap, = ctx.saved_tensors
        dZ2 = dZ2.contiguous()
        dT = torch.empty((B, Hm, Wm, ldt), dtype=torch.float32, device=dZ2.device)
        rc = rows_ctx(ctx.step, B)
        check(hip.lib().sp_sal_gather_bwd_rows(ptr(dZ2), B, Hm, Wm, ldt, nsel, nsrc, ptr(hmap), ptr(dT), ptr(rc.last) if rc is not None else None,
                                               int(ctx.step) if rc is not None else 0, hip.stream()), "sp_sal_gather_bwd_rows")
        return dT, None, None, None, None


def sal_gather(T, hmap, nsel, nsrc, step=None):
    return _SalGather.apply(T, hmap, nsel, nsrc, step)


class _DrtDirect(Function):
    """h [B,Hm,Wm,C], W11 [nheads,ncls,121,C], cbsum [nheads,ncls] -> Dpre [nsel,B,dh*dw]"""
    @staticmethod
    def forward(ctx, h, W11, cbsum, hmap, nsel, step=None, batch=None):
        # batch (DrtBatch): the forward values of ALL decode steps are computed in one launch behind the loop (drt_heads_batched): this
        # node then only reserves the step's output and keeps what its backward -- which stays a per-step launch inside the backward
        # recurrence, beside the h-gate conv's data gradient -- needs
        ctx.step, ctx.batch = step, batch
        h = h.contiguous()
        W11 = W11.contiguous()
        cbsum = cbsum.contiguous()
        B, Hm, Wm, C_ = h.shape
        S = ((Hm + 4 - 7) // 5 + 1) * ((Wm + 4 - 7) // 5 + 1)
        if batch is None:
            Dpre = torch.empty((nsel, B, S), dtype=torch.float32, device=h.device)
            check(hip.lib().sp_drt_direct_fwd(ptr(h), ptr(W11), ptr(cbsum), ptr(hmap), B, Hm, Wm, C_, nsel, ptr(Dpre), hip.stream()),
                  "sp_drt_direct_fwd")
        else:
            ctx.slot, Dpre = batch.add(h, W11, cbsum, hmap, nsel, S)          # this step's slice of the batch's [T, nsel, B, S] buffer
        ctx.cfg = (B, Hm, Wm, C_, nsel, W11.shape[0], tuple(W11.shape), tuple(cbsum.shape))
        ctx.save_for_backward(h, W11, hmap)
        return Dpre

    @staticmethod
    def backward(ctx, dD):
        B, Hm, Wm, C_, nsel, nheads, wshape, cshape = ctx.cfg
        h, W11, hmap = ctx.saved_tensors
        dD = dD.contiguous()
        L = hip.lib()
        dh = dW = dcs = None
        rc = rows_ctx(ctx.step, B)      # samples behind their last loss step: exactly-zero dD -> nothing computed for them, h not read
        # (slot, sample) pairs whose duration gradient is exactly zero -- AiR's unselected head, the step at a scanpath's end -- as the
        # batched duration backward found them (drt_heads_batched): skipped like the dead samples, bit-identical sums
        live = ctx.batch.live_of(ctx.slot) if (ctx.batch is not None and ROW_SPARSITY) else None
        rl, rs = (ptr(rc.last), int(ctx.step)) if rc is not None else (None, 0)
        if ctx.needs_input_grad[0]:
            dh = torch.empty_like(h)
            check(L.sp_drt_direct_bwd_data_live(ptr(dD), ptr(W11), ptr(hmap), B, Hm, Wm, C_, nsel, 0, ptr(dh), ptr(live), rl, rs, B,
                                                hip.stream()), "sp_drt_direct_bwd_data_live")
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dW = torch.empty(wshape, dtype=torch.float32, device=h.device)
            dcs = torch.empty(cshape, dtype=torch.float32, device=h.device)
            ws = hip.workspace(L.sp_drt_direct_bwd_weight_workspace(B, Hm, Wm, C_, nsel), h.device, slot=0)
            check(L.sp_drt_direct_bwd_weight_live(ptr(dD), ptr(h), ptr(hmap), B, Hm, Wm, C_, nsel, nheads, ptr(ws), ptr(dW), ptr(dcs),
                                                  ptr(live), rl, rs, B, hip.stream()), "sp_drt_direct_bwd_weight_live")
        return dh, dW, dcs, None, None, None, None


def drt_direct(h, W11, cbsum, hmap, nsel, step=None, batch=None):
    return _DrtDirect.apply(h, W11, cbsum, hmap, nsel, step, batch)


class DrtBatch:
    """The duration sites of ALL decode steps in one forward launch (round 6).  Nothing in the recurrence reads the duration branch of
    predict_head (AiR/models/baseline_attention.py:155-159: mu and sigma2 only leave the model), but per step its window kernel stood in
    the serial gap between two fused h-gate launches (112 us of 468, L2-bound).  The decode loop's drt_direct(..., batch=this) nodes only
    register their operands; drt_heads_batched() then runs sp_drt_direct_fwd ONCE over the T x B hidden states -- contiguous when the cell
    kernels wrote them into hbuf() -- and the duration part of the head epilogue once over T x nheads virtual heads.  Backward: the batched
    node turns d(mu), d(sigma2) into the site gradients of every step (one launch) and into `live` flags; the per-step nodes' data / weight
    gradient launches stay where they were, inside the backward recurrence beside the h-gate conv's data gradient."""

    def __init__(self, T):
        self.T, self.items, self.live, self.buf, self.D = T, [], None, None, None

    def hbuf(self, t, like):
        """the slot of decode step t in ONE [T, B, Hm, Wm, C] buffer for the hidden states (cell kernels write h_t there)"""
        if self.buf is None:
            self.buf = torch.empty((self.T,) + tuple(like.shape), dtype=torch.float32, device=like.device)
        return self.buf[t]

    def add(self, h, W11, cbsum, hmap, nsel, S):
        """register decode step len(items); returns (slot, its [nsel, B, S] slice of the site buffer drt_heads_batched fills)"""
        if self.D is None:
            self.D = torch.empty((self.T, nsel, h.shape[0], S), dtype=torch.float32, device=h.device)
        slot = len(self.items)
        if slot >= self.T:
            raise RuntimeError("scanpaths_amd: DrtBatch got more decode steps than it was built for")
        self.items.append((h, W11, cbsum, hmap, nsel))
        return slot, self.D[slot]

    def live_of(self, slot):
        return None if self.live is None else self.live[slot]


class _DrtHeadsBatched(Function):
    """placeholders Dpre_t [nsel,B,S] of the T per-step drt_direct nodes (graph edges; their contents are produced HERE) + cb, w2, b2 ->
    mu, sigma2 [nsel, B, T]"""
    @staticmethod
    def forward(ctx, batch, cb, w2, b2, HC, per_sample, *dpres):
        items = batch.items
        Tn = len(items)
        assert Tn == len(dpres) and Tn >= 1
        h0, W11, cbsum, hmap, nsel = items[0]
        B, Hm, Wm, C_ = h0.shape
        S = dpres[0].shape[-1]
        dev = h0.device
        L = hip.lib()
        hs = [it[0] for it in items]
        nb = h0.numel() * 4
        contiguous = all(h.is_contiguous() and h.data_ptr() == hs[0].data_ptr() + t * nb for t, h in enumerate(hs))
        same_w = all(it[1].data_ptr() == W11.data_ptr() and it[2].data_ptr() == cbsum.data_ptr() for it in items)
        D = batch.D[:Tn]          # [T, nsel, B, S]: the per-step nodes' outputs are its slices
        if contiguous and same_w and Tn > 1:
            Dall = torch.empty((nsel, Tn * B, S), dtype=torch.float32, device=dev)
            hm = hmap.repeat(Tn, 1).contiguous()
            check(L.sp_drt_direct_fwd(ptr(hs[0]), ptr(W11), ptr(cbsum), ptr(hm), Tn * B, Hm, Wm, C_, nsel, ptr(Dall), hip.stream()),
                  "sp_drt_direct_fwd (batched)")
            D.copy_(Dall.view(nsel, Tn, B, S).permute(1, 0, 2, 3))
            FUSION_COUNTS["drt_fwd_batched"] += 1
        else:          # (hidden states that were not written into DrtBatch.hbuf: one launch per step, still behind the loop)
            for t, (h, w11, cs, hm_, _) in enumerate(items):
                check(L.sp_drt_direct_fwd(ptr(h.contiguous()), ptr(w11), ptr(cs), ptr(hm_), B, Hm, Wm, C_, nsel, ptr(D[t]), hip.stream()),
                      "sp_drt_direct_fwd")
        cb = cb.contiguous()
        w2c = w2.detach().contiguous()
        b2 = b2.contiguous()
        NH = Tn * nsel          # virtual heads (t, i): head (t, i) uses cb[.., i], shared w2 / b2
        cb_all = (cb.view(B, nsel, HC).repeat(1, Tn, 1) if per_sample else cb.view(nsel, HC).repeat(Tn, 1)).contiguous()
        mu = torch.empty((NH, B), dtype=torch.float32, device=dev)
        s2 = torch.empty((NH, B), dtype=torch.float32, device=dev)
        drt = torch.empty((NH, B, S), dtype=torch.float32, device=dev)
        check(L.sp_head_finish_parts_fwd(None, B, Hm, Wm, 0, NH, HC, ptr(cb_all), int(per_sample), ptr(w2c), ptr(b2), 0, None, None, ptr(mu),
                                         ptr(s2), ptr(drt), ptr(D), 2, 2, hip.stream()), "sp_head_finish_parts_fwd (duration)")
        ctx.batch = batch
        ctx.cfg = (Tn, nsel, B, Hm, Wm, S, HC, per_sample, tuple(w2.shape), tuple(cb.shape))
        ctx.save_for_backward(s2, drt, w2c)
        to_bt = lambda v: v.view(Tn, nsel, B).permute(1, 2, 0).contiguous()          # [nsel, B, T]
        return to_bt(mu), to_bt(s2)

    @staticmethod
    def backward(ctx, dmu, ds2):
        Tn, nsel, B, Hm, Wm, S, HC, per_sample, w2shape, cbshape = ctx.cfg
        s2, drt, w2c = ctx.saved_tensors
        dev = s2.device
        NH = Tn * nsel
        to_tb = lambda v, ref: (v.permute(2, 0, 1).contiguous().view(NH, B) if v is not None else torch.zeros_like(ref))
        dmu, ds2 = to_tb(dmu, s2), to_tb(ds2, s2)
        ddpre = torch.empty((NH, B, S), dtype=torch.float32, device=dev)
        dcbp = torch.empty((B, NH * HC), dtype=torch.float32, device=dev)
        dw2p = torch.empty((B, NH * 2 * S), dtype=torch.float32, device=dev)
        db2p = torch.empty((B, NH * 2), dtype=torch.float32, device=dev)
        live = torch.empty((NH * B,), dtype=torch.int32, device=dev)
        check(hip.lib().sp_head_finish_parts_bwd(None, None, ptr(dmu), ptr(ds2), None, None, ptr(s2), ptr(drt), B, Hm, Wm, 0, NH, HC, ptr(w2c), 0,
                                                 None, ptr(dcbp), ptr(dw2p), ptr(db2p), ptr(ddpre), 2, 2, ptr(live), hip.stream()),
              "sp_head_finish_parts_bwd (duration)")
        ctx.batch.live = live.view(Tn, nsel * B)          # row t: the flags [nsel][B] of decode step t's drt_direct node
        # partial sums over the samples (and the T steps): composed bias (only drt_layer_1.bias' entry is non-zero), drt_layer_2
        if per_sample:
            dcb = dcbp.view(B, Tn, nsel * HC).sum(1).view(cbshape)
        else:
            dcb = _colsum_any(dcbp, NH * HC).view(Tn, nsel * HC).sum(0).view(cbshape)
        dw2 = _colsum_any(dw2p, NH * 2 * S).view(NH, 2 * S).sum(0)
        db2 = _colsum_any(db2p, NH * 2).view(NH, 2).sum(0)
        dd = ddpre.view(Tn, nsel, B, S)
        return (None, dcb, dw2.reshape(w2shape), db2.reshape(2), None, None) + tuple(dd[t] for t in range(Tn))


def drt_heads_batched(batch: DrtBatch, dpres, cb, w2, b2, HC, per_sample=False):
    """mu, sigma2 [nsel, B, T] of all decode steps registered with `batch` (see DrtBatch)"""
    return _DrtHeadsBatched.apply(batch, cb, w2, b2, HC, per_sample, *dpres)


# ----------------------------------------------------------------------------------------------------
# loss
# ----------------------------------------------------------------------------------------------------
def device_sum(x: torch.Tensor) -> torch.Tensor:
    x = x.contiguous()
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    L = hip.lib()
    ws = hip.workspace(L.sp_sumsq_workspace(x.numel()), x.device, slot=1)
    check(L.sp_sum(ptr(x), x.numel(), ptr(out), ptr(ws), hip.stream()), "sp_sum")
    return out


class _ScanpathLoss(Function):
    @staticmethod
    def forward(ctx, z, mu, sigma2, gt, amask, dur, dmask, lambda1, mask_sums):
        z, mu, sigma2 = z.contiguous(), mu.contiguous(), sigma2.contiguous()
        gt, amask, dur, dmask = gt.contiguous(), amask.contiguous(), dur.contiguous(), dmask.contiguous()
        B, T, A = z.shape
        dev = z.device
        L = hip.lib()
        out3 = torch.empty(3, dtype=torch.float32, device=dev)
        dz = torch.empty_like(z)
        dmu = torch.empty_like(mu)
        ds2 = torch.empty_like(sigma2)
        ws = hip.workspace(L.sp_scanpath_loss_workspace(B, T), dev, slot=1)
        check(L.sp_scanpath_loss(ptr(z), ptr(gt), ptr(amask), ptr(mu), ptr(sigma2), ptr(dur), ptr(dmask), B, T, A,
                                 float(lambda1), ptr(mask_sums), ptr(out3), ptr(dz), ptr(dmu), ptr(ds2), ptr(ws),
                                 hip.stream()), "sp_scanpath_loss")
        ctx.save_for_backward(dz, dmu, ds2)
        loss, la, ld = out3[0].clone(), out3[1].clone(), out3[2].clone()
        ctx.mark_non_differentiable(la, ld)
        return loss, la, ld

    @staticmethod
    def backward(ctx, g, _ga, _gd):
        dz, dmu, ds2 = ctx.saved_tensors
        g = g.reshape(1).contiguous().to(torch.float32)
        L = hip.lib()
        outs = []
        for t in (dz, dmu, ds2):
            o = torch.empty_like(t)
            check(L.sp_scale_by(ptr(t), ptr(g), t.numel(), ptr(o), hip.stream()), "sp_scale_by")
            outs.append(o)
        return outs[0], outs[1], outs[2], None, None, None, None, None, None


def scanpath_loss(z, mu, sigma2, gt, amask, dur, dmask, lambda1=1.0, mask_sums=None):
    """loss, loss_actions, loss_duration (AiR/train.py:192-197).  mask_sums: device tensor
    [sum(action_masks), sum(duration_masks)]; computed locally when None (single-process semantics).
    (The gradient rows of masked-out steps are exact zeros; the decoder's backward pass skips what they imply by itself, see _OutputGate.)"""
    if mask_sums is None:
        mask_sums = torch.cat([device_sum(amask), device_sum(dmask)])
    return _ScanpathLoss.apply(z, mu, sigma2, gt, amask, dur, dmask, lambda1, mask_sums)
