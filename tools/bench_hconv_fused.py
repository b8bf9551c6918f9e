"""The three dominant launches of the bench line exactly as the training step issues them (bs 32, 40x64 map, C = 512):
   * h2_kernel<fwd, LSTM epilogue>  (sp_gateconv_lstm_f16x2: h-gate conv + ConvLSTM cell, 15 of 16 forward launches per step),
   * h2_kernel<dgrad>               (data gradient of the h-gate conv from the split dpre the cell backward wrote),
   * hw_kernel                      (its weight gradient),
   plus lstm_bwd_kernel between them.  HIP-event timed; profile with
   rocprofv3 --pmc <counters> --output-format csv -d <dir> -o p -- python3 tools/bench_hconv_fused.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from scanpaths_amd import functional as F, hip

B, Hm, Wm, C, KP = 32, 40, 64, 512, 20
P = Hm * Wm
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
h0 = (torch.randn(B, Hm, Wm, C, generator=g) * torch.rand(B, Hm, Wm, C, generator=g)).to(dev)
c0 = torch.randn(B, Hm, Wm, C, generator=g).to(dev)
xg = torch.randn(B, Hm, Wm, 4 * C, generator=g).to(dev).requires_grad_(True)
w = (torch.randn(4 * C, C, 3, 3, generator=g) * 0.02).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
spcol = torch.rand(B, P, KP, generator=g).to(dev)
wc = (torch.randn(B, 3 * C, KP, generator=g) * 0.1).to(dev)
gh = torch.randn(B, Hm, Wm, C, generator=g).to(dev)
gh._sp_amax = None


def step():
    h = h0.clone().requires_grad_(True)
    c = c0.clone().requires_grad_(True)
    c._sp_cbound = 4.0
    hn, cn = F.gateconv_lstm(h, w, xg, c, spcol, wc, {})
    # gradients arrive with their max|.| hints in the model (fan-in pass): reproduce that so the cell backward writes the split dpre
    gg = gh.clone()
    hint = torch.zeros(2, device=dev)          # its own tensor: an in-place torch write into a POOL slot would bump the version counter
    hint[1] = float(gg.abs().max())            # every saved pool slot shares (autograd then refuses the saved operand scales)
    gg._sp_amax = hint
    torch.autograd.backward([hn], [gg])


for _ in range(3):
    step()
F.reset_fusion_counts()
hip.TIMER = hip.KernelTimer(min_flops=1e9)
for _ in range(int(os.environ.get("N_ITER", "10"))):
    step()
torch.cuda.synchronize()
out = {k[0]: {"avg_ms": round(d["avg_ms"], 4), "tflops": round(d["tflops"], 1), "launches": d["launches"]} for k, d in hip.TIMER.summary().items()}
out["fusion_counts"] = {k: v for k, v in F.FUSION_COUNTS.items() if v}
print(json.dumps(out))
