"""A/B timing of the workgroup order ("hw_map") and split-K count ("hw_splits") of the 2xfp16-split weight-gradient kernel
(csrc/conv_f16x2.hip hw_kernel) on the dominant shape: h-gate conv 3x3 512->2048 at bs 32, 40x64 (M = 81 920 pixels, Co = 2048,
(tap, ci) = 4608).  Interleaved rounds in one process; equal split counts must give bit-identical results whatever the order.
    python3 tools/bench_hw_map.py [rounds] [reps]        env CONFIGS="0:0,1:0,0:16,1:16" (map:splits, splits 0 = heuristic)"""
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
gy = (torch.randn(B, Hm, Wm, 4 * C, generator=g) * 1e-3).to(dev)
L = hip.lib()
hs, gys = F.split_op(h, "f16x2"), F.split_op(gy, "f16x2")
dw = torch.empty(4 * C, 3, 3, C, device=dev)
FL = 2.0 * B * Hm * Wm * 4 * C * 9 * C


def wgrad():
    F._wgrad_b3(hs, gys, dw, N_img=B, Hi=Hm, Wi=Wm, Ci=C, Ho=Hm, Wo=Wm, Co=4 * C, ldo=9 * C, KH=3, KW=3, pad=1)


def setup(cfg):
    hip.check(L.sp_set_tuning(b"hw_map", cfg[0]), "sp_set_tuning")
    hip.check(L.sp_set_tuning(b"hw_splits", cfg[1] if cfg[1] > 0 else -1), "sp_set_tuning")


def timed():
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(REPS):
        wgrad()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / REPS


cfgs = [tuple(int(v) for v in c.split(":")) for c in os.environ.get("CONFIGS", "0:0,1:0,0:16,1:16,1:24,0:6,1:6").split(",")]
outs, times = {}, {c: [] for c in cfgs}
for c in cfgs:
    setup(c)
    wgrad()
    torch.cuda.synchronize()
    outs[c] = dw.clone()
for _ in range(ROUNDS):
    for c in cfgs:
        setup(c)
        times[c].append(timed())
ref = outs[cfgs[0]]
res = {}
for c in cfgs:
    same = [o for o in cfgs if o[1] == c[1] and o != c]
    res[f"map{c[0]}_splits{c[1]}"] = {
        "median_ms": round(statistics.median(times[c]), 4), "min_ms": round(min(times[c]), 4),
        "tflops_median": round(FL / statistics.median(times[c]) / 1e9, 1),
        "rel_diff_to_first": float((outs[c] - ref).abs().max() / ref.abs().max()),
        "bit_identical_to_same_splits": all(bool(torch.equal(outs[c], outs[o])) for o in same)}
print(json.dumps({"shape": "M=81920 Co=2048 (tap,ci)=4608", "rounds": ROUNDS, "reps": REPS, "configs": res}))
