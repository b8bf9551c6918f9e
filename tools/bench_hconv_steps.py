"""The dominant launches of the bench line as the training step issues them since round 4 (bs 32, 40x64 map, C = 512, T decode steps):
   * h2_kernel<fwd, LSTM epilogue>  (sp_gateconv_lstm_f16x2), T - 1 launches,
   * h2_kernel<dgrad>               (data gradient of the h-gate conv from the split gate gradient), T - 1 launches,
   * hw2_kernel                     (the weight gradient of ALL T - 1 applications in ONE launch at the end of BPTT) + its slab reduce,
   with lstm_bwd_kernel between them.  HIP-event timed; profile with
   rocprofv3 --pmc <counters> --output-format csv -d <dir> -o p -- python3 tools/bench_hconv_steps.py       (tools/run_r04_pmc.sh)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from scanpaths_amd import functional as F, hip

B, Hm, Wm, C, KP = 32, 40, 64, 512, 20
T = int(os.environ.get("T_STEPS", "16"))
P = Hm * Wm
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
h0 = (torch.randn(B, Hm, Wm, C, generator=g) * torch.rand(B, Hm, Wm, C, generator=g)).to(dev)
c0 = torch.randn(B, Hm, Wm, C, generator=g).to(dev)
xg = torch.randn(B, Hm, Wm, 4 * C, generator=g).to(dev).requires_grad_(True)
w = (torch.randn(4 * C, C, 3, 3, generator=g) * 0.02).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
spcol = torch.rand(B, P, KP, generator=g).to(dev)
wc = (torch.randn(B, 3 * C, KP, generator=g) * 0.1).to(dev)
gh = (torch.randn(B, Hm, Wm, C, generator=g) * 0.05).to(dev)


def step():
    cache = {"defer": F.DeferredWgrad()}
    h = h0.clone().requires_grad_(True)
    c = c0.clone().requires_grad_(True)
    c._sp_cbound = 1.0
    xgs = F.fanout(xg, T - 1)
    loss = 0.0
    for t in range(T - 1):
        h, c = F.gateconv_lstm(h, w, xgs[t], c, spcol, wc, cache)
        hs = F.fanout(h, 2)                       # a head consumer and the next step, as in the decoder (gives dh its max|.| hint)
        loss = loss + (hs[0] * gh).sum()
        h = hs[1]
    loss.backward()


for _ in range(2):
    step()
F.reset_fusion_counts()
hip.TIMER = hip.KernelTimer(min_flops=1e9)
for _ in range(int(os.environ.get("N_ITER", "3"))):
    step()
torch.cuda.synchronize()
out = {k[0]: {"avg_ms": round(d["avg_ms"], 4), "tflops": round(d["tflops"], 1), "launches": d["launches"], "M": k[1]}
       for k, d in hip.TIMER.summary().items()}
out["fusion_counts"] = {k: v for k, v in F.FUSION_COUNTS.items() if v}
print(json.dumps(out))
