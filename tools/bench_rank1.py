#!/usr/bin/env python3
"""The two gradients of the rank-1 gate term at the benchmark shape (B 32, P 2560, 3C 1536, KP 20), alone on the GPU: the fused kernel
(csrc/rank1_grads.hip) against the two batched split GEMMs it replaces, with and without their operand preparation, on a full batch and
on batches where only 24 / 16 / 8 samples still carry loss gradient (row_last).  One JSON line (us per call)."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from scanpaths_amd import functional as F, hip
    from scanpaths_amd.hip import ConvDesc
    dev, L = torch.device("cuda", 0), hip.lib()
    B, P, Cc, KP = 32, 2560, 512, 20
    C4, N3 = 4 * Cc, 3 * Cc
    g = torch.Generator(device="cpu").manual_seed(0)
    dpre = torch.randn(B, P, C4, generator=g).to(dev)
    spcol, wc = torch.randn(B, P, KP, generator=g).to(dev), (torch.randn(B, N3, KP, generator=g) * 0.2).to(dev)
    ys = F.split_op(dpre)
    wsp = torch.empty(L.sp_rank1_grads_workspace(B, P, N3, KP), dtype=torch.uint8, device=dev)
    dsp, dwc = torch.empty_like(spcol), torch.empty_like(wc)
    ptr, st = hip.ptr, hip.stream()

    def fused(prep=True):
        ws = F.split_w(wc.transpose(1, 2).contiguous().view(B * KP, N3), "f16x2")
        F.check(L.sp_rank1_grads_f16x2(ptr(ys.buf), ptr(ys.scale), C4, ptr(ws.buf), ptr(ws.scale), ptr(spcol), B, P, N3, KP, ptr(dsp), ptr(dwc),
                                       ptr(wsp), None, 0, st), "fused")

    xs0 = F.split_op(torch.nn.functional.pad(spcol, (0, 32 - KP)), "f16x2")
    ws0 = F.split_w(wc.transpose(1, 2).contiguous().view(B * KP, N3), "f16x2")

    def fused_kernel_only():
        F.check(L.sp_rank1_grads_f16x2(ptr(ys.buf), ptr(ys.scale), C4, ptr(ws0.buf), ptr(ws0.scale), ptr(spcol), B, P, N3, KP, ptr(dsp), ptr(dwc),
                                       ptr(wsp), None, 0, st), "fused")

    def pair():
        ws = F.split_w(wc.transpose(1, 2).contiguous().view(B * KP, N3), "f16x2")
        d = ConvDesc(P, 1, 1, N3, C4, 1, 1, KP, KP, 1, 1, 1, 0, 1, 0, N3, 1.0, 0, 0, B, P * C4, KP * N3, P * KP, 0, None)
        d.w_scale_rows = 1
        F.check(L.sp_conv_igemm_f16x2(C.byref(d), ptr(ys.buf), ptr(ys.scale), ptr(ws.buf), ptr(ws.scale), None, ptr(dsp), st), "igemm")
        xs = F.split_op(torch.nn.functional.pad(spcol, (0, 32 - KP)), "f16x2")
        d2 = hip.WgradDesc(1, P // 64, 64, 32, 32, P // 64, 64, N3, C4, 1, 1, 1, 0, 1, KP, 0, 1.0, B, P * 32, P * C4, N3 * KP)
        F.check(L.sp_conv_wgrad_f16x2(C.byref(d2), ptr(xs.buf), ptr(xs.scale), ptr(ys.buf), ptr(ys.scale), ptr(dwc), None, st), "wgrad")

    def timeit(f, n=30):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            f()
        e1.record()
        torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) / n * 1e3, 1)
    out = {"shape": dict(B=B, P=P, N3=N3, KP=KP), "fused_with_prep_us": timeit(fused), "fused_kernel_us": timeit(fused_kernel_only),
           "two_gemms_with_prep_us": timeit(pair), "planes_bytes_read_once": 2 * 2 * B * P * N3}
    for live in (24, 16, 8):
        last = torch.tensor([9] * live + [0] * (B - live), dtype=torch.int32, device=dev)
        def fk():
            F.check(L.sp_rank1_grads_f16x2(ptr(ys.buf), ptr(ys.scale), C4, ptr(ws0.buf), ptr(ws0.scale), ptr(spcol), B, P, N3, KP, ptr(dsp), ptr(dwc),
                                           ptr(wsp), ptr(last), 5, st), "fused")
        def pr():
            d = ConvDesc(P, 1, 1, N3, C4, 1, 1, KP, KP, 1, 1, 1, 0, 1, 0, N3, 1.0, 0, 0, B, P * C4, KP * N3, P * KP, 0, None)
            d.w_scale_rows = 1
            d.row_last, d.row_step = last.data_ptr(), 5
            F.check(L.sp_conv_igemm_f16x2(C.byref(d), ptr(ys.buf), ptr(ys.scale), ptr(ws0.buf), ptr(ws0.scale), None, ptr(dsp), st), "igemm")
            d2 = hip.WgradDesc(1, P // 64, 64, 32, 32, P // 64, 64, N3, C4, 1, 1, 1, 0, 1, KP, 0, 1.0, B, P * 32, P * C4, N3 * KP)
            d2.row_last, d2.row_step = last.data_ptr(), 5
            F.check(L.sp_conv_wgrad_f16x2(C.byref(d2), ptr(xs0.buf), ptr(xs0.scale), ptr(ys.buf), ptr(ys.scale), ptr(dwc), None, st), "wgrad")
        out["live%d_fused_kernel_us" % live] = timeit(fk)
        out["live%d_two_gemms_no_prep_us" % live] = timeit(pr)
    out["fused_kernel_GBps"] = round(out["planes_bytes_read_once"] / out["fused_kernel_us"] / 1e3, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
