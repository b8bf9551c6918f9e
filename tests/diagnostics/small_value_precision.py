"""Diagnostic (GPU box): relative error of the 2xfp16-split GEMMs on operand entries that are SMALL relative to the tensor maximum.
One entry of the activation operand is 1 (it sets the per-tensor scale), all others are u * 2^-e with u in [1, 2).  Design claim
(csrc/conv_f16x2.hip header): absolute error 2^-39 * amax, i.e. relative 2^(e - 39 + 14) -- unless the matrix pipe flushes fp16
subnormals (the residual plane of entries below 2^-16.5 * amax is subnormal), in which case the error jumps to 2^-11.
    python tests/diagnostics/small_value_precision.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from scanpaths_amd import functional as F      # noqa: E402

dev = torch.device("cuda:0")
F._b3_pays = lambda *a, **k: True
F._w3_pays = lambda *a, **k: True
M, K, N = 8192, 512, 128
g = torch.Generator().manual_seed(3)
w = torch.randn(N, K, 1, 1, generator=g)
rows = []
for scheme in ("f16x2", "bf16x3"):
    F.SPLIT_SCHEME = scheme
    for e in (0, 4, 8, 12, 14, 16, 17, 18, 20, 22, 24, 26):
        x = (1 + torch.rand(1, M // 64, 64, K, generator=g)) * 2.0 ** -e * torch.where(torch.rand(1, M // 64, 64, K, generator=g) < 0.5, -1.0, 1.0)
        x[0, 0, 0, 0] = 1.0
        xg = x.to(dev).requires_grad_(True)
        wg = w.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        y = F.conv2d(xg, wg, None)
        gy = torch.randn(y.shape, generator=g).to(dev)
        y.backward(gy)
        ref = x.double().reshape(M, K) @ w.double().reshape(N, K).t()
        got = y.detach().cpu().double().reshape(M, N)
        err = ((got - ref)[64:].pow(2).mean().sqrt() / ref[64:].pow(2).mean().sqrt()).item()      # rows that do not contain the 1
        refw = gy.cpu().double().reshape(M, N).t() @ x.double().reshape(M, K)
        gotw = wg.grad.detach().cpu().double().reshape(N, K)
        errw = ((gotw - refw)[:, 1:].pow(2).mean().sqrt() / refw[:, 1:].pow(2).mean().sqrt()).item()
        rows.append({"scheme": scheme, "e": e, "fwd_rel_rms": err, "wgrad_rel_rms": errw})
        print(f"{scheme}: entries 2^-{e:<2d} of the maximum: forward rel rms err {err:.2e}   weight gradient {errw:.2e}", flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out", "diag"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "diag", "small_value_precision.json"), "w"), indent=0)
