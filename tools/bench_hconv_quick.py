"""h-gate conv (M = 81920, N = 2048, K = 4608) forward / data gradient / weight gradient on the default back-end only, HIP-event timed."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from scanpaths_amd import functional as F, hip

B, Hm, Wm, C = 32, 40, 64, 512
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
h = torch.randn(B, Hm, Wm, C, generator=g).to(dev).requires_grad_(True)
w = (torch.randn(4 * C, C, 3, 3, generator=g) * 0.02).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
gy = torch.randn(B, Hm, Wm, 4 * C, generator=g).to(dev)
for _ in range(3):
    y = F.conv2d(h, w, None, pad=1)
    y.backward(gy)
hip.TIMER = hip.KernelTimer(min_flops=1e9)
for _ in range(int(os.environ.get("N_ITER", "12"))):
    y = F.conv2d(h, w, None, pad=1)
    y.backward(gy)
torch.cuda.synchronize()
print(json.dumps({k[0]: {"avg_ms": round(d["avg_ms"], 4), "tflops": round(d["tflops"], 1)} for k, d in hip.TIMER.summary().items()}))
