"""Error of the three GEMM back-ends (fp32 MFMA, 3xbf16 split, 2xfp16 split) against fp64 on decoder-shaped data:
   the h-gate conv (h in (-1,1) with many near-zero gates) and its dgrad/wgrad.  Prints rms error / rms(result)."""
import os, sys, json, math
import torch
import torch.nn.functional as TF
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scanpaths_amd import functional as F

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, Hm, Wm, C = 2, 30, 40, 512
o = torch.sigmoid(torch.randn(B, C, Hm, Wm, generator=g) * 2)
c = torch.randn(B, C, Hm, Wm, generator=g) * 0.7
h = (o * c)                                                   # ConvLSTM h = o * c
w = torch.randn(4 * C, C, 3, 3, generator=g) * (1.0 / math.sqrt(9 * C))
gy = torch.randn(B, 4 * C, Hm, Wm, generator=g) * torch.rand(B, 4 * C, Hm, Wm, generator=g) ** 4      # heavy-tailed gradient
hr, wr = h.double().requires_grad_(True), w.double().requires_grad_(True)
yr = TF.conv2d(hr, wr, padding=1)
yr.backward(gy.double())
h32, w32 = h.clone().requires_grad_(True), w.clone().requires_grad_(True)
y32 = TF.conv2d(h32, w32, padding=1)
y32.backward(gy)
rel = lambda a, b: ((a.double().cpu() - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()
out = {"torch_cpu_fp32": {"y": rel(y32.detach(), yr.detach()), "dh": rel(h32.grad, hr.grad), "dw": rel(w32.grad, wr.grad)}}
force = lambda *a, **k: True
F._b3_pays, F._w3_pays = force, force
for name, use, scheme in (("fp32_mfma", False, "bf16x3"), ("bf16x3", True, "bf16x3"), ("f16x2", True, "f16x2")):
    F.USE_BF16X3, F.SPLIT_SCHEME = use, scheme
    if not use:
        F._b3_pays, F._w3_pays = (lambda *a, **k: False), (lambda *a, **k: False)
    else:
        F._b3_pays, F._w3_pays = force, force
    hd = h.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    wd = w.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = F.conv2d(hd, wd, None, pad=1)
    y.backward(gy.permute(0, 2, 3, 1).contiguous().to(dev))
    out[name] = {"y": rel(y.detach().permute(0, 3, 1, 2), yr.detach()), "dh": rel(hd.grad.permute(0, 3, 1, 2), hr.grad),
                 "dw": rel(wd.grad, wr.grad)}
print(json.dumps(out, indent=1))
