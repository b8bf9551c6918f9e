"""Micro-benchmark of the dominant kernel only: the per-step h-gate conv (implicit GEMM M=B*P, N=2048, K=4608) and
its dgrad / wgrad, timed with HIP events on the launch stream.  Profile with
   rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_hconv -- python3 tools/bench_hconv.py"""
import os, sys, json
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from scanpaths_amd import functional as F, hip

B, Hm, Wm, C = 32, 40, 64, 512
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
h = torch.randn(B, Hm, Wm, C, generator=g).to(dev).requires_grad_(True)
w = (torch.randn(4 * C, C, 3, 3, generator=g) * 0.02).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
gy = torch.randn(B, Hm, Wm, 4 * C, generator=g).to(dev)
for _ in range(2):
    y = F.conv2d(h, w, None, pad=1); y.backward(gy)
hip.TIMER = hip.KernelTimer(min_flops=1e9)
for _ in range(10):
    y = F.conv2d(h, w, None, pad=1); y.backward(gy)
torch.cuda.synchronize()
if F.SPLIT_SCHEME != "bf16x3":          # the 3 x bf16 / 6-product scheme next to the default 2 x fp16 / 3-product one
    F.SPLIT_SCHEME = "bf16x3"
    for _ in range(6):
        y = F.conv2d(h, w, None, pad=1); y.backward(gy)
    torch.cuda.synchronize()
out = {}
F.USE_BF16X3 = False
for _ in range(6):
    y = F.conv2d(h, w, None, pad=1); y.backward(gy)
torch.cuda.synchronize()
for k, d in hip.TIMER.summary().items():
    out[k[0]] = {"avg_ms": round(d["avg_ms"], 4), "tflops": round(d["tflops"], 2), "launches": d["launches"],
                 "frac_of_157.3": round(d["tflops"] / 157.3, 4)}
print(json.dumps(out))
