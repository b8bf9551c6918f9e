"""Timing of the encoder's short-K pointwise convolutions (1x1, bs 32) on the 2xfp16 path: forward with / without the BatchNorm
statistics epilogue, and the data gradient.   python3 tools/bench_pointwise.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from scanpaths_amd import functional as F  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(80, 128, 128, 512), (80, 128, 256, 64), (80, 128, 64, 256), (40, 64, 256, 1024), (40, 64, 1024, 256), (40, 64, 512, 2048)]
REPS = 20


def timed(fn):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(REPS):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / REPS * 1e3


out = []
for H, W, Ci, Co in SHAPES:
    x = torch.randn(32, H, W, Ci, device=dev)
    w = (torch.randn(Co, Ci, 1, 1, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    x._sp_cache = {}
    F.split_op(x)                                           # the operand is there, as after a BatchNorm that emitted it
    with torch.no_grad():
        t_plain = timed(lambda: F.conv2d(x, w, None))
        t_stats = timed(lambda: F.conv2d(x, w, None, bn_stats=True))
    gb = (x.numel() + 32 * H * W * Co) * 4 / 1e9
    out.append({"shape": f"M={32 * H * W} K={Ci} N={Co}", "fwd_us": round(t_plain, 1), "fwd_stats_us": round(t_stats, 1),
                "GB": round(gb, 3), "TBps_plain": round(gb / t_plain * 1e3, 2)})
    print(out[-1], flush=True)
print(json.dumps(out))
