"""Does the row stride of the split operands (power of two: 8 KB per pixel for 2048 channels) matter?  Same GEMM kernels on
channel counts next to 512 / 2048: algorithmic TFLOP/s per shape.   python3 tools/bench_stride.py"""
import json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from scanpaths_amd import functional as F, hip

B, Hm, Wm = 32, 40, 64
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
out = {}
for Ci, Co in ((512, 2048), (512, 2112), (544, 2048), (544, 2112), (512, 1984), (480, 2048)):
    h = (torch.randn(B, Hm, Wm, Ci, generator=g) * torch.rand(B, Hm, Wm, Ci, generator=g)).to(dev)
    wp = (torch.randn(Co, 3, 3, Ci, generator=g) * 0.02).to(dev)
    gy = (torch.randn(B, Hm, Wm, Co, generator=g) * 1e-3).to(dev)
    hs, ws, wT, gys = F.split_op(h, "f16x2"), F.split_op(wp, "f16x2"), F.split_op_wT(wp, "f16x2"), F.split_op(gy, "f16x2")
    y = torch.empty(B, Hm, Wm, Co, device=dev); dx = torch.empty(B, Hm, Wm, Ci, device=dev); dw = torch.empty_like(wp)
    FL = 2.0 * B * Hm * Wm * Co * 9 * Ci
    fns = {"fwd": lambda: F._igemm_b3(hs, ws, None, y, N_img=B, Hi=Hm, Wi=Wm, Kc=Ci, ldx=Ci, Ho=Hm, Wo=Wm, Nout=Co, ldc=Co, ldw=9 * Ci, KH=3, KW=3, pad=1, mode=0),
           "dgrad": lambda: F._igemm_b3(gys, wT, None, dx, N_img=B, Hi=Hm, Wi=Wm, Kc=Co, ldx=Co, Ho=Hm, Wo=Wm, Nout=Ci, ldc=Ci, ldw=9 * Co, KH=3, KW=3, pad=1, mode=1),
           "wgrad": lambda: F._wgrad_b3(hs, gys, dw, N_img=B, Hi=Hm, Wi=Wm, Ci=Ci, Ho=Hm, Wo=Wm, Co=Co, ldo=9 * Ci, KH=3, KW=3, pad=1)}
    res = {}
    for name, fn in fns.items():
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            fn()
        e.record(); e.synchronize()
        ms = s.elapsed_time(e) / 10
        res[name] = {"ms": round(ms, 3), "tflops": round(FL / ms / 1e9, 1)}
    out[f"Ci={Ci},Co={Co}"] = res
    del h, wp, gy, hs, ws, wT, gys, y, dx, dw
    torch.cuda.empty_cache()
print(json.dumps(out))
