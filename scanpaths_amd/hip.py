"""ctypes binding of libscanpaths_amd.so (the C ABI declared in include/scanpaths_amd.h).

The product path has NO CPU fallback: if the shared library is missing or a tensor is not on a HIP
device, the calls raise.  torch is used only for device memory and the current stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import config as _config

_HERE = os.path.dirname(os.path.abspath(__file__))
# config library="timing" (tools/ only; SP_LIBRARY=timing under SP_ALLOW_ENV_TUNING=1): the A/B build with wrong-result timing modes
# compiled in (make -C scanpaths_amd/csrc timing).  The product path always loads libscanpaths_amd.so.
TIMING_LIB = False
LIB_PATH = ""


def _apply_config():
    global TIMING_LIB, LIB_PATH
    TIMING_LIB = _config.settings["library"] == "timing"
    LIB_PATH = os.path.join(_HERE, "libscanpaths_amd_timing.so" if TIMING_LIB else "libscanpaths_amd.so")


_apply_config()


class ConvDesc(C.Structure):
    _fields_ = [("N_img", C.c_int), ("Hi", C.c_int), ("Wi", C.c_int), ("Kc", C.c_int), ("ldx", C.c_int),
                ("Ho", C.c_int), ("Wo", C.c_int), ("Nout", C.c_int), ("ldc", C.c_int),
                ("KH", C.c_int), ("KW", C.c_int), ("stride", C.c_int), ("pad", C.c_int), ("dil", C.c_int),
                ("mode", C.c_int), ("ldw", C.c_int),
                ("alpha", C.c_float), ("beta", C.c_int), ("relu", C.c_int),
                ("nbatch", C.c_int),
                ("strideX", C.c_int64), ("strideW", C.c_int64), ("strideC", C.c_int64),
                ("ksplit", C.c_int), ("workspace", C.c_void_p), ("w_scale_rows", C.c_int),
                ("row_last", C.c_void_p), ("row_step", C.c_int)]


class WgradDesc(C.Structure):
    _fields_ = [("N_img", C.c_int), ("Hi", C.c_int), ("Wi", C.c_int), ("Ci", C.c_int), ("ldx", C.c_int),
                ("Ho", C.c_int), ("Wo", C.c_int), ("Co", C.c_int), ("ldy", C.c_int),
                ("KH", C.c_int), ("KW", C.c_int), ("stride", C.c_int), ("pad", C.c_int), ("dil", C.c_int),
                ("ldo", C.c_int), ("beta", C.c_int), ("alpha", C.c_float), ("nbatch", C.c_int),
                ("strideX", C.c_int64), ("strideY", C.c_int64), ("strideO", C.c_int64),
                ("x_scale_vec", C.c_int), ("y_scale_vec", C.c_int), ("row_last", C.c_void_p), ("row_step", C.c_int)]


_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
ABI_VERSION = 3      # = SP_ABI_VERSION of include/scanpaths_amd.h (held equal by tests/test_cpu_host.py)

# name -> (restype, argtypes); must list every symbol include/scanpaths_amd.h declares
SIGNATURES = {
    "sp_abi_version": (_I, []),
    "sp_set_tuning": (_I, [C.c_char_p, _I]),
    "sp_timing_build": (_I, []),
    "sp_conv_igemm": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P]),
    "sp_split3_bf16": (_I, [_P, _L, _P, _P]),
    "sp_split3_bf16_wT": (_I, [_P, _I, _I, _I, _P, _P]),
    "sp_split2_f16": (_I, [_P, _L, _P, _P, _I, _P]),
    "sp_split2_f16_wT": (_I, [_P, _I, _I, _I, _P, _P, _P]),
    "sp_split2_f16_rows": (_I, [_P, _L, _L, _I, _P, _P, _P, _P]),
    "sp_split2_f16_wT_rows": (_I, [_P, _I, _I, _I, _P, _P, _P, _P]),
    "sp_split2_f16_wT_rows_batched": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "sp_split2_f16_cols_workspace": (_L, [_L, _I]),
    "sp_split2_f16_cols": (_I, [_P, _L, _I, _P, _P, _P, _P]),
    "sp_conv_wgrad_f16x2_multi_workspace": (_L, [_P, _I]),
    "sp_conv_wgrad_f16x2_multi": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sp_conv_igemm_f16x2": (_I, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "sp_conv_wgrad_f16x2_workspace": (_L, [_P]),
    "sp_conv_wgrad_f16x2": (_I, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "sp_conv_igemm_f16x1": (_I, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "sp_conv_wgrad_f16x1": (_I, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "sp_gateconv_lstm_f16x2": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _F, _P]),
    "sp_rank1_grads_applies": (_I, [_I, _I, _I, _I, _I]),
    "sp_rank1_grads_workspace": (_L, [_I, _I, _I, _I]),
    "sp_rank1_grads_f16x2": (_I, [_P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P]),
    "sp_conv_igemm_bf16x3": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P]),
    "sp_conv_wgrad_bf16x3_workspace": (_L, [C.POINTER(WgradDesc)]),
    "sp_conv_wgrad_bf16x3": (_I, [C.POINTER(WgradDesc), _P, _P, _P, _P, _P]),
    "sp_conv_wgrad_workspace": (_L, [C.POINTER(WgradDesc)]),
    "sp_conv_wgrad": (_I, [C.POINTER(WgradDesc), _P, _P, _P, _P, _P]),
    "sp_colsum_workspace": (_L, [_L, _I]),
    "sp_colsum": (_I, [_P, _L, _I, _I, _P, _I, _P, _P]),
    "sp_rowsum": (_I, [_P, _L, _I, _F, _P, _P]),
    "sp_rowsum_bwd": (_I, [_P, _L, _I, _F, _P, _P]),
    "sp_bn_workspace": (_L, [_L, _I]),
    "sp_bn_stats": (_I, [_P, _L, _I, _F, _F, _P, _P, _P, _P, _P, _P]),
    "sp_bn_eval_stats": (_I, [_P, _P, _I, _F, _P, _P, _P]),
    "sp_bn_apply": (_I, [_P, _P, _P, _P, _P, _P, _I, _L, _I, _P, _P, _P]),
    "sp_bn_backward": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _L, _I, _P, _P, _P, _P, _P, _P, _P]),
    "sp_bn_split_workspace": (_L, [_L, _I]),
    "sp_bn_mask_words": (_L, [_L, _I]),
    "sp_bn_fwd_split": (_I, [_P, _L, _I, _F, _F, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "sp_conv_stats_tiles": (_L, [C.POINTER(ConvDesc)]),
    "sp_conv_igemm_f16x2_stats": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    "sp_bn_bwd_split": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sp_sum_n": (_I, [_P, _I, _L, _P, _P, _P]),
    "sp_sum_n_rows": (_I, [_P, _I, _L, _P, _P, _P, _P, _I, _P]),
    "sp_sum_n_mixed": (_I, [_P, _P, _P, _I, _L, _P, _P, _P]),
    "sp_sum_n_mixed_rows": (_I, [_P, _P, _P, _I, _L, _P, _P, _P, _P, _I, _P]),
    "sp_relu_bwd": (_I, [_P, _P, _L, _P, _P]),
    "sp_maxpool3s2_fwd": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _P]),
    "sp_maxpool3s2_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _I, _I, _P]),
    "sp_maxpool3s2_fwd_idx": (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _I, _P]),
    "sp_maxpool3s2_bwd_idx": (_I, [_P, _P, _I, _I, _I, _I, _P, _I, _I, _P]),
    "sp_nchw_to_nhwc_pad": (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "sp_pad_lastdim": (_I, [_P, _L, _I, _I, _P, _P]),
    "sp_add": (_I, [_P, _P, _P, _L, _P]),
    "sp_lstm_pointwise_fwd": (_I, [_P, _P, _P, _L, _I, _P, _P, _P, _P]),
    "sp_lstm_rank1_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "sp_rank1_dwc_workspace": (_L, [_I, _I, _I, _I]),
    "sp_rank1_dwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "sp_sempool_workspace": (_L, [_I, _I, _I, _I]),
    "sp_sempool_fwd": (_I, [_P, _P, _I, _I, _I, _I, _F, _P, _P, _P]),
    "sp_sempool_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P]),
    "sp_sempool_bwd_rows": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P, _I, _P]),
    "sp_sempool_fwd_sbc": (_I, [_P, _P, _I, _I, _I, _I, _F, _P, _P, _I, _P]),
    "sp_sempool_bwd_rows_sbc": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P, _I, _I, _P]),
    "sp_lstm_pointwise_bwd": (_I, [_P, _P, _P, _P, _P, _L, _I, _P, _P, _P, _P]),
    "sp_lstm_pointwise_bwd_split": (_I, [_P, _P, _P, _P, _P, _L, _I, _P, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P]),
    "sp_lstm_pointwise_bwd_rows": (_I, [_P, _P, _P, _P, _P, _L, _I, _P, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _I, _I, _P]),
    "sp_im2col3x3_1ch": (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "sp_col2im3x3_1ch": (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "sp_im2col3x3_multi": (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "sp_col2im3x3_multi": (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "sp_listatt_fwd": (_I, [_P, _P, _I, _I, _I, _P, _P, _P]),
    "sp_listatt_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "sp_mulrelu_fwd": (_I, [_P, _P, _L, _L, _P, _P]),
    "sp_mulrelu_bwd": (_I, [_P, _P, _P, _P, _L, _L, _P, _P, _P]),
    "sp_select_rows": (_I, [_P, _P, _P, _L, _L, _P, _P]),
    "sp_select_rows_bwd": (_I, [_P, _P, _L, _L, _P, _P, _P]),
    "sp_head_finish_fwd": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P]),
    "sp_head_finish_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _I,
                                _P]),
    "sp_head_finish_parts_fwd": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "sp_head_finish_parts_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _I,
                                      _I, _P, _P]),
    "sp_head_finish_parts_bwd_ld": (_I, [_P, _L, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _I,
                                      _I, _P, _P]),
    "sp_head_num_classes": (_I, [_I, _I]),
    "sp_head_compose11_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "sp_head_compose11_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "sp_sal_gather_fwd": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "sp_sal_gather_bwd": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "sp_sal_gather_bwd_rows": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P]),
    "sp_drt_direct_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "sp_drt_direct_bwd_data": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "sp_drt_direct_bwd_data_live": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _I, _P]),
    "sp_drt_direct_bwd_weight_live": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _I, _P]),
    "sp_drt_direct_bwd_weight_workspace": (_L, [_I, _I, _I, _I, _I]),
    "sp_drt_direct_bwd_weight": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "sp_drt_direct_bwd_weight_rows": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P]),
    "sp_scanmatch_max_len": (_I, []),
    "sp_scanmatch_submatrix": (_I, [_I, _I, C.c_double, _P, _P, _P]),
    "sp_scanmatch_sequences": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _I, C.c_double, C.c_double, C.c_double, _P, _I, _P, _P, _P]),
    "sp_scanmatch_score": (_I, [_P, _P, _I, _P, _P, _I, _P, _I, _P, _I, _P, C.c_double, _P, _P]),
    "sp_scanmatch_score_long_workspace": (_L, [_I, _I]),
    "sp_scanmatch_score_long": (_I, [_P, _P, _I, _P, _P, _I, _P, _I, _P, _I, _P, C.c_double, _P, _P, _P]),
    "sp_scanmatch_align": (_I, [_P, _I, _P, _I, _P, _I, _P, C.c_double, _P, _P, _P, _P, _P, _P]),
    "sp_scan_max_fixations": (_I, []),
    "sp_scan_sed_stde": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, C.c_double, _P, _P, _P]),
    "sp_scan_multimatch": (_I, [_P, _I, _P, _P, _P, _I, C.c_double, C.c_double, _P, _P]),
    "sp_sample_actions": (_I, [_P, _P, _P, _I, _I, _I, _I, C.c_uint64, _P, _P, _P, _P]),
    "sp_generate_scanpath": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "sp_beam_search": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "sp_collate_targets": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "sp_blur_targets": (_I, [_P, _I, _I, _I, C.c_double, _P]),
    "sp_scanpath_loss_workspace": (_L, [_I, _I]),
    "sp_scanpath_loss": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P, _P]),
    "sp_log_action": (_I, [_P, _P, _I, _I, _P, _P, _P, _P]),
    "sp_log_duration": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P]),
    "sp_rowscale": (_I, [_P, _P, _I, _I, _P, _P]),
    "sp_rows_last": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "sp_gemm_skinny_applies": (_I, [_I, _I, _I, _I, _I, _I, _I]),
    "sp_gemm_skinny_workspace": (_L, [_I, _I, _I, _I]),
    "sp_gemm_skinny": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _I, _P, _P]),
    "sp_scale_by": (_I, [_P, _P, _L, _P, _P]),
    "sp_sumsq_workspace": (_L, [_L]),
    "sp_sum": (_I, [_P, _L, _P, _P, _P]),
    "sp_sumsq": (_I, [_P, _L, _P, _P, _P]),
    "sp_clip_adam": (_I, [_P, _P, _P, _P, _L, _P, _F, _F, _F, _F, _F, _F, _F, _F, _F, _P]),
}

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load the HIP library (once).  Raises loudly when it has not been built: there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  scanpaths_amd has no CPU or eager fallback.")
        _lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args
        if _lib.sp_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH}: ABI version {_lib.sp_abi_version()}, this binding expects {ABI_VERSION} (SP_ABI_VERSION in "
                               "include/scanpaths_amd.h; bumped whenever a descriptor layout or an entry point's meaning changes): rebuild "
                               "with `make -C scanpaths_amd/csrc`")
        # every max|.| slot this host passes comes zeroed from functional._amax_hint's pool and is used once: the launchers add
        # their reset node only while a stream is being captured (graph replays re-use the slot)
        if not _config.settings["always_reset_amax"]:
            check(_lib.sp_set_tuning(b"amax_reset", 1), "sp_set_tuning")
        if _lib.sp_timing_build() != int(TIMING_LIB):
            raise RuntimeError(f"{LIB_PATH}: timing / product build mix-up")
        if TIMING_LIB and _config.env_tuning_allowed():      # A/B timing / profiling knobs, honoured by the timing build only
            for env, knob in _config.TIMING_KNOBS.items():
                if os.environ.get(env):
                    check(_lib.sp_set_tuning(knob, int(os.environ[env])), "sp_set_tuning")
    return _lib


class HipError(RuntimeError):
    pass


def check(rc: int, what: str) -> None:
    if rc != 0:
        kind = {-1: "SP_EINVAL (unsupported shape/alignment)", -2: "SP_ENULL (null pointer)"}.get(rc, f"hipError {rc}")
        raise HipError(f"{what} failed: {kind}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise HipError("scanpaths_amd kernels need tensors on a HIP device (no CPU path)")
    return t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


_side = {}


def side_stream(device) -> "torch.cuda.Stream":
    """one extra HIP stream per device: in the backward recurrence it carries the h-gate conv's data gradient of decode step t
    (functional._GateConvLstm.backward) beside the current stream's chain of small launches; the fan-in of h_{t-1}'s gradients
    (functional._FanOut.backward) waits for the event that launch leaves behind.  Deferred weight gradients run on the CURRENT stream."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    s = _side.get(key)
    if s is None:
        # lowest priority the device offers (this pool's MI355X boxes: range (0, -1), i.e. normal): the side stream carries ONE long GEMM
        # beside the current stream's chain of small launches (A/B of normal vs high: 233.5 / 233.7 vs 233.6 / 234.6 ms per step)
        try:
            lo = max(torch.cuda.Stream.priority_range())
        except Exception:
            lo = 0
        s = torch.cuda.Stream(device=device, priority=lo)
        _side[key] = s
    return s


_ws = {}


def workspace(nbytes: int, device, slot: int = 0) -> Optional[torch.Tensor]:
    """Grow-only scratch buffer per (device, stream, slot); safe because the kernels of one stream run in order and a slot is consumed by
    the launch that requested it before the next request on that stream (round 5: one data gradient per decode step runs on the side
    stream beside the current one -- each stream has its own scratch)."""
    if nbytes <= 0:
        return None
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream, slot)
    cur = _ws.get(key)
    if cur is None or cur.numel() < nbytes:
        cur = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws[key] = cur
    return cur


class KernelTimer:
    """HIP-event timing of individual GEMM launches on the launch stream (bench.py roofline leg).
    Only launches with at least ``min_flops`` algorithmic FLOPs are bracketed, so the timed region is not perturbed
    by ~2000 extra event records per step."""

    def __init__(self, min_flops: float = 1e11):
        self.min_flops = min_flops
        self.pending = []          # (key, flops, start_event, end_event)

    def bracket(self, key, flops, launch):
        if flops < self.min_flops:
            return launch()
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        launch()
        e.record()
        # a launch enqueued on the side stream runs BESIDE the current stream's small launches: its time includes that contention and is
        # not comparable with a stand-alone launch of the same kernel -- the key says so ("+side"), bench.py reports it as `beside_chain`
        cur = torch.cuda.current_stream()
        sd = _side.get(cur.device.index if cur.device.index is not None else torch.cuda.current_device())
        if sd is not None and cur.cuda_stream == sd.cuda_stream and isinstance(key, tuple) and key and isinstance(key[0], str):
            key = (key[0] + "+side",) + tuple(key[1:])
        self.pending.append((key, flops, s, e))

    def summary(self):
        """key -> dict(launches, flops_per_launch, avg_ms, tflops); call after a device synchronise."""
        out = {}
        for key, flops, s, e in self.pending:
            d = out.setdefault(key, {"launches": 0, "flops_per_launch": flops, "ms": 0.0})
            d["launches"] += 1
            d["ms"] += s.elapsed_time(e)
        for d in out.values():
            d["avg_ms"] = d["ms"] / d["launches"]
            d["tflops"] = d["flops_per_launch"] / (d["avg_ms"] * 1e-3) / 1e12
        return out


TIMER: Optional[KernelTimer] = None
