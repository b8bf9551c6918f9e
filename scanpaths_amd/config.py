"""One switchboard for the arithmetic back-end, the library build and the fusion switches of the package (VERDICT r3 "next" #8).

Set in CODE:

    from scanpaths_amd import config
    config.set(split_scheme="bf16x3")          # before or after scanpaths_amd.functional is imported

Environment variables are honoured ONLY when ``SP_ALLOW_ENV_TUNING=1`` is set as well (the A/B tools under tools/ and the
diagnostics under tests/diagnostics do that); without it a set ``SP_*`` tuning variable is IGNORED and says so once on stderr --
a stray variable in a job's environment must not be able to move a training run onto a narrower arithmetic or a wrong-result
timing library.  Anything that leaves the fp32-faithful default arithmetic (``split_scheme="f16x1"``) or the product library
(``library="timing"``) prints one loud line when it is selected, and ``config.non_default()`` is what bench.py tags its JSON with.

switch            env variable (needs SP_ALLOW_ENV_TUNING=1)   meaning
split_scheme      SP_SPLIT_SCHEME = f16x2 | bf16x3 | f16x1      GEMM back-end: 2 x fp16 planes / 3 products (default), 3 x bf16 planes /
                                                               6 products, or THROUGHPUT MODE (one fp16 plane, fails the 1e-4 parity bar)
use_split         SP_NO_SPLIT=1 -> False                        False: every GEMM on the fp32 MFMA kernels
library           SP_LIBRARY = timing                           "timing": libscanpaths_amd_timing.so (wrong-result timing modes; tools only)
fused_amax, grad_merge, bn_split, bn_skip_dx, bn_skip_z, lstm_bwd_split, rank1_dsp_split, rank1_dwc_split, lstm_skip_dpre,
fuse_gate_lstm, lstm_h_planes, defer_wgrad, channel_scales, hw2_single, row_sparsity, direct_grad, skinny_gemm, async_dgrad, drt_batched, rank1_fused      fusion switches (SP_NO_AMAX_HINT, SP_GRAD_MERGE, SP_BN_SPLIT, SP_BN_SKIP_DX,
                                                               SP_BN_SKIP_Z, SP_LSTM_BWD_SPLIT, SP_RANK1_DSP_SPLIT, SP_RANK1_DWC_SPLIT,
                                                               SP_LSTM_SKIP_DPRE, SP_FUSE_LSTM, SP_LSTM_H_PLANES, SP_DEFER_WGRAD,
                                                               SP_CHANNEL_SCALES, SP_HW2_SINGLE, SP_ROW_SPARSITY, SP_DIRECT_GRAD, SP_SKINNY_GEMM, SP_ASYNC_DGRAD, SP_DRT_BATCHED, SP_RANK1_FUSED: "0" switches off); all default on, results agree to rounding
"""
from __future__ import annotations

import os
import sys

DEFAULTS = {
    "split_scheme": "f16x2", "use_split": True, "library": "product", "always_reset_amax": False,
    "fused_amax": True, "grad_merge": True, "bn_split": True, "bn_skip_dx": True, "bn_skip_z": True, "lstm_bwd_split": True,
    "rank1_dsp_split": True, "rank1_dwc_split": True, "lstm_skip_dpre": True, "fuse_gate_lstm": True, "lstm_h_planes": True,
    "defer_wgrad": True, "channel_scales": True, "hw2_single": True, "row_sparsity": True, "direct_grad": True, "skinny_gemm": True, "async_dgrad": True,
    "drt_batched": True, "rank1_fused": True,
}
_off = lambda v: v != "0"
_ENV = {       # env name -> (switch, parser)
    "SP_SPLIT_SCHEME": ("split_scheme", str), "SP_NO_SPLIT": ("use_split", lambda v: not v), "SP_LIBRARY": ("library", str),
    "SP_ALWAYS_RESET_AMAX": ("always_reset_amax", bool), "SP_NO_AMAX_HINT": ("fused_amax", lambda v: not v),
    "SP_GRAD_MERGE": ("grad_merge", _off), "SP_BN_SPLIT": ("bn_split", _off), "SP_BN_SKIP_DX": ("bn_skip_dx", _off),
    "SP_BN_SKIP_Z": ("bn_skip_z", _off), "SP_LSTM_BWD_SPLIT": ("lstm_bwd_split", _off), "SP_RANK1_DSP_SPLIT": ("rank1_dsp_split", _off),
    "SP_RANK1_DWC_SPLIT": ("rank1_dwc_split", _off), "SP_LSTM_SKIP_DPRE": ("lstm_skip_dpre", _off), "SP_FUSE_LSTM": ("fuse_gate_lstm", _off),
    "SP_LSTM_H_PLANES": ("lstm_h_planes", _off), "SP_DEFER_WGRAD": ("defer_wgrad", _off), "SP_CHANNEL_SCALES": ("channel_scales", _off),
    "SP_HW2_SINGLE": ("hw2_single", _off), "SP_ROW_SPARSITY": ("row_sparsity", _off), "SP_DIRECT_GRAD": ("direct_grad", _off), "SP_SKINNY_GEMM": ("skinny_gemm", _off),
    "SP_ASYNC_DGRAD": ("async_dgrad", _off), "SP_DRT_BATCHED": ("drt_batched", _off), "SP_RANK1_FUSED": ("rank1_fused", _off),
}
# timing-library knobs (sp_set_tuning; honoured by libscanpaths_amd_timing.so only): passed through under SP_ALLOW_ENV_TUNING=1
TIMING_KNOBS = {"SP_H2_DBG": b"h2_dbg", "SP_HW_DBG": b"hw_dbg", "SP_B3_DBG": b"b3_dbg", "SP_HW_SPLITS": b"hw_splits", "SP_H2_HALO": b"h2_halo",
                "SP_ROW_ORDER": b"row_order", "SP_HW_CAP": b"hw_cap"}

settings = dict(DEFAULTS)
_announced = set()


def env_tuning_allowed() -> bool:
    return os.environ.get("SP_ALLOW_ENV_TUNING") == "1"


def _loud(msg: str) -> None:
    if msg not in _announced:
        _announced.add(msg)
        print(f"scanpaths_amd.config: {msg}", file=sys.stderr, flush=True)


def _check(name, value):
    if name not in DEFAULTS:
        raise KeyError(f"scanpaths_amd.config: unknown switch {name!r} (known: {sorted(DEFAULTS)})")
    if name == "split_scheme" and value not in ("f16x2", "bf16x3", "f16x1"):
        raise ValueError(f"split_scheme must be f16x2, bf16x3 or f16x1, got {value!r}")
    if name == "library" and value not in ("product", "timing"):
        raise ValueError(f"library must be product or timing, got {value!r}")
    if name == "split_scheme" and value == "f16x1":
        _loud("THROUGHPUT MODE selected (split_scheme=f16x1): GEMM operands rounded to ONE fp16 plane -- about 100x off the 1e-4 parity "
              "bar; results are NOT comparable with the fp32 reference")
    if name == "library" and value == "timing":
        _loud("TIMING LIBRARY selected (libscanpaths_amd_timing.so): contains wrong-result timing modes; for tools/ only")


def _load_env():
    present = [k for k in list(_ENV) + list(TIMING_KNOBS) if os.environ.get(k) not in (None, "")]
    if not present:
        return
    if not env_tuning_allowed():
        _loud(f"IGNORING {', '.join(present)}: environment tuning needs SP_ALLOW_ENV_TUNING=1 (set switches in code with "
              "scanpaths_amd.config.set)")
        return
    for k in present:
        if k in _ENV:
            name, parse = _ENV[k]
            v = parse(os.environ[k])
            _check(name, v)
            settings[name] = v


def set(**kw) -> None:      # noqa: A001 (deliberately the module's verb)
    """change switches; takes effect immediately in scanpaths_amd.functional if it is already imported (the library choice must be made
    before the first kernel launch)"""
    for k, v in kw.items():
        _check(k, v)
        if k == "library" and "scanpaths_amd.hip" in sys.modules and sys.modules["scanpaths_amd.hip"]._lib is not None \
                and v != settings["library"]:
            raise RuntimeError("scanpaths_amd.config: the library is already loaded; choose it before the first launch")
        settings[k] = v
    f = sys.modules.get("scanpaths_amd.functional")
    if f is not None:
        f._apply_config()
    h = sys.modules.get("scanpaths_amd.hip")
    if h is not None and "library" in kw:
        h._apply_config()


def honour_env_for_tools() -> None:
    """For tools/ and tests/diagnostics only -- scripts whose PURPOSE is to A/B a switch named on their command line
    (`SP_SPLIT_SCHEME=f16x1 python3 tools/precision_mode_error.py`, `SP_LIBRARY=timing python3 tools/bench_drt.py`): equivalent to
    exporting SP_ALLOW_ENV_TUNING=1 before the import.  Call it before the first kernel launch (the library choice is fixed there).
    Values go through the same checks and loud lines as config.set(); the product path never calls this."""
    os.environ["SP_ALLOW_ENV_TUNING"] = "1"
    _load_env()
    for name in ("scanpaths_amd.functional", "scanpaths_amd.hip"):
        m = sys.modules.get(name)
        if m is not None:
            if name.endswith("hip") and m._lib is not None and (settings["library"] == "timing") != m.TIMING_LIB:
                raise RuntimeError("scanpaths_amd.config: the library is already loaded; call honour_env_for_tools() before the first launch")
            m._apply_config()


def non_default() -> dict:
    """the switches that differ from the defaults (bench.py tags its JSON line with them)"""
    return {k: v for k, v in settings.items() if v != DEFAULTS[k]}


_load_env()
