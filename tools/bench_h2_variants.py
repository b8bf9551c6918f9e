"""A/B timing of the schedule variants of the 2xfp16-split GEMM kernels (csrc/conv_f16x2.hip: h2_kernel VAR 0..3, hw_kernel
VAR 0/2/3) on the dominant shape: h-gate conv 3x3 512->2048 at bs 32, 40x64 (M = 81 920, N = 2048, K = 4608).
Interleaved rounds in ONE process (cdna_hip_programming.md rule 24): every round times every variant once (REPS launches
between two HIP events on the launch stream); prints median and min per variant and checks that all variants produce
bit-identical outputs.      python3 tools/bench_h2_variants.py [rounds] [reps]"""
import json
import os
os.environ.setdefault("SP_LIBRARY", "timing")      # schedule variants / timing modes live in libscanpaths_amd_timing.so (make timing)
import statistics
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from scanpaths_amd import functional as F, hip  # noqa: E402

ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 5
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B, Hm, Wm, C = 32, 40, 64, 512
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
h = (torch.randn(B, Hm, Wm, C, generator=g) * torch.rand(B, Hm, Wm, C, generator=g)).to(dev)
wp = (torch.randn(4 * C, 3, 3, C, generator=g) * 0.02).to(dev)                     # physical [Co][KH][KW][Ci]
gy = (torch.randn(B, Hm, Wm, 4 * C, generator=g) * 1e-3).to(dev)
L = hip.lib()
hs, ws, wT, gys = F.split_op(h, "f16x2"), F.split_op(wp, "f16x2"), F.split_op_wT(wp, "f16x2"), F.split_op(gy, "f16x2")
y = torch.empty(B, Hm, Wm, 4 * C, device=dev)
dx = torch.empty(B, Hm, Wm, C, device=dev)
dw = torch.empty_like(wp)
FL = 2.0 * B * Hm * Wm * 4 * C * 9 * C


def fwd():
    F._igemm_b3(hs, ws, None, y, N_img=B, Hi=Hm, Wi=Wm, Kc=C, ldx=C, Ho=Hm, Wo=Wm, Nout=4 * C, ldc=4 * C, ldw=9 * C, KH=3, KW=3, pad=1, mode=0)


def dgrad():
    F._igemm_b3(gys, wT, None, dx, N_img=B, Hi=Hm, Wi=Wm, Kc=4 * C, ldx=4 * C, Ho=Hm, Wo=Wm, Nout=C, ldc=C, ldw=9 * 4 * C, KH=3, KW=3, pad=1, mode=1)


def wgrad():
    F._wgrad_b3(hs, gys, dw, N_img=B, Hi=Hm, Wi=Wm, Ci=C, Ho=Hm, Wo=Wm, Co=4 * C, ldo=9 * C, KH=3, KW=3, pad=1)


def timed(fn):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(REPS):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / REPS


H2V = [int(v) for v in os.environ.get("H2_VARIANTS", "0,3,4,7,8,11,12,14,15").split(",")]      # 4..7 = the 16x16x32 MFMA shape
cases = [("fwd", fwd, b"h2_variant", H2V, lambda: y), ("dgrad", dgrad, b"h2_variant", H2V, lambda: dx),
         ("wgrad", wgrad, b"hw_variant", [int(v) for v in os.environ.get("HW_VARIANTS", "0,2,6,7").split(",")], lambda: dw)]
if os.environ.get("ONLY"):
    cases = [c for c in cases if c[0] in os.environ["ONLY"].split(",")]
res, ident = {}, {}
for name, fn, knob, variants, outp in cases:
    ref = None
    for v in variants:                       # warm-up + bit-identity
        hip.check(L.sp_set_tuning(knob, v), "sp_set_tuning")
        fn()
        torch.cuda.synchronize()
        if ref is None:
            ref = outp().clone()
        ident[f"{name}_v{v}"] = bool(torch.equal(ref, outp())) if v < 4 else \
            f"rel max diff {float((ref - outp()).abs().max() / ref.abs().max()):.2e}"
    times = {v: [] for v in variants}
    for _ in range(ROUNDS):
        for v in variants:
            hip.check(L.sp_set_tuning(knob, v), "sp_set_tuning")
            times[v].append(timed(fn))
    hip.check(L.sp_set_tuning(knob, -1), "sp_set_tuning")
    res[name] = {f"v{v}": {"median_ms": round(statistics.median(t), 4), "min_ms": round(min(t), 4),
                           "tflops_median": round(FL / statistics.median(t) / 1e9, 1)} for v, t in times.items()}
print(json.dumps({"shape": "M=81920 N=2048 K=4608", "rounds": ROUNDS, "reps": REPS, "ms": res, "bit_identical_to_v0": ident}))
