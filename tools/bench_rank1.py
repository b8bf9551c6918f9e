"""The cell backward's two rank-1 gradients at the benchmark size (B 32, P 2560, 3C = 1536 of 4C = 2048 channels, KP 20):
spatial-tap gradient through the fp32-MFMA batched GEMM vs the batched split GEMM; the filter gradient kernel.   python3 tools/bench_rank1.py"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from scanpaths_amd import functional as F, hip  # noqa: E402

dev = torch.device("cuda:0")
B, P, N3, C4, KP = 32, 2560, 1536, 2048, 20
g = torch.Generator().manual_seed(0)
dpre = torch.randn(B * P, C4, generator=g).to(dev)
wc = (torch.randn(B, N3, KP, generator=g) * 0.05).to(dev)
spcol = torch.randn(B, P, KP, generator=g).to(dev)
xs = F.split_op(dpre, "f16x2")
L = hip.lib()


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return round(s.elapsed_time(e) / reps * 1e3, 1)


dsp0, dsp1 = torch.empty_like(spcol), torch.empty_like(spcol)


def fp32_path():
    F._igemm(dpre, wc, None, dsp0, N_img=P, Hi=1, Wi=1, Kc=N3, ldx=C4, Ho=1, Wo=1, Nout=KP, ldc=KP, ldw=KP, mode=1, nbatch=B, sX=P * C4, sW=N3 * KP,
             sC=P * KP)


def prep():
    return F.split_op(wc.transpose(1, 2).contiguous(), "f16x2")


ws = prep()
d = hip.ConvDesc(P, 1, 1, N3, C4, 1, 1, KP, KP, 1, 1, 1, 0, 1, 0, N3, 1.0, 0, 0, B, P * C4, KP * N3, P * KP, 0, None)


def split_gemm():
    hip.check(L.sp_conv_igemm_f16x2(C.byref(d), hip.ptr(xs.buf), hip.ptr(xs.scale), hip.ptr(ws.buf), hip.ptr(ws.scale), None, hip.ptr(dsp1), hip.stream()), "b")


res = {"fp32_batched_us": timed(fp32_path), "split_prep_us": timed(prep), "split_gemm_us": timed(split_gemm)}
torch.cuda.synchronize()
ref = torch.einsum("bpk,bkn->bpn", dpre.view(B, P, C4)[:, :, :N3].double().cpu(), wc.double().cpu())
res["err_fp32"] = float((dsp0.double().cpu() - ref).abs().max() / ref.abs().max())
res["err_split"] = float((dsp1.double().cpu() - ref).abs().max() / ref.abs().max())
dwc = torch.empty_like(wc)
wsb = hip.workspace(L.sp_rank1_dwc_workspace(B, P, N3, KP), dev, slot=0)
res["rank1_dwc_us"] = timed(lambda: hip.check(L.sp_rank1_dwc(hip.ptr(dpre), hip.ptr(spcol), B, P, C4, N3, KP, hip.ptr(wsb), hip.ptr(dwc), hip.stream()), "r"))
ys = xs
dwc1 = torch.empty_like(wc)


def dwc_prep():
    return F.split_op(torch.nn.functional.pad(spcol, (0, 32 - KP)), "f16x2")


xsp = dwc_prep()
dw = hip.WgradDesc(1, P // 64, 64, 32, 32, P // 64, 64, N3, C4, 1, 1, 1, 0, 1, KP, 0, 1.0, B, P * 32, P * C4, N3 * KP)
res["dwc_split_prep_us"] = timed(dwc_prep)
res["dwc_split_gemm_us"] = timed(lambda: hip.check(L.sp_conv_wgrad_f16x2(C.byref(dw), hip.ptr(xsp.buf), hip.ptr(xsp.scale), hip.ptr(ys.buf), hip.ptr(ys.scale),
                                                                      hip.ptr(dwc1), None, hip.stream()), "w"))
torch.cuda.synchronize()
refw = torch.einsum("bpc,bpk->bck", dpre.view(B, P, C4)[:, :, :N3].double().cpu(), spcol.double().cpu())
res["err_dwc_valu"] = float((dwc.double().cpu() - refw).abs().max() / refw.abs().max())
res["err_dwc_split"] = float((dwc1.double().cpu() - refw).abs().max() / refw.abs().max())
print(json.dumps(res))
