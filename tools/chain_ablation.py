#!/usr/bin/env python3
"""What each piece of the backward recurrence's chain of small launches costs the TRAINING STEP (wrong results: timing only).

The chain (backward of the memory update and of the previous step's heads) runs on the current stream beside the h-gate conv's data
gradient on the side stream; its kernels' durations in a trace are inflated by that co-residence and say little about what removing or
shrinking one of them would buy.  This tool measures it directly: the bench step (bs 32, 320x512, T = 16) with one piece at a time
replaced by a no-op that hands autograd tensors of the right shape (uninitialised memory -- the numbers are garbage, the launches that
remain are real), same process, interleaved with the unmodified step.

    python3 tools/chain_ablation.py [--steps 6] [--rounds 2]  ->  one JSON line (ms per step per variant, delta vs base)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--rounds", type=int, default=2)
    args = ap.parse_args()
    from scanpaths_amd import functional as F
    from scanpaths_amd.models.baseline_attention import baseline
    from scanpaths_amd.models.loss import supervised_loss
    from scanpaths_amd.optim import FlatAdam
    from scanpaths_amd.procedural import fill_module
    from scanpaths_amd.synth import make_batch
    dev = torch.device("cuda", 0)
    T = 16
    model = baseline(convLSTM_length=T, map_width=64, map_height=40)
    fill_module(model, seed=0)
    model = model.to(dev).train()
    b = {k: v.to(dev) for k, v in make_batch("AiR", 32, 320, 512, T, seed=0).items()}
    opt = FlatAdam(model.parameters(), lr=1e-4, weight_decay=5e-5, clip=12.5)

    def step():
        opt.zero_grad()
        pred = model(b["images"], b["attention_maps"], b["performances"])
        loss, _, _ = supervised_loss(pred, b["scanpaths"], b["durations"], b["action_masks"], b["duration_masks"], 1.0)
        loss.backward()
        opt.step()

    def timed():
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps * 1e3

    # ---- ablations: (name, apply() -> undo()) --------------------------------------------------------------------------------------
    def patch(cls, fn):
        old = cls.backward
        cls.backward = staticmethod(fn)
        return lambda: setattr(cls, "backward", old)

    def no_drt_bwd():
        def bwd(ctx, dD):
            B, Hm, Wm, C_, nsel, nheads, wshape, cshape = ctx.cfg
            h, W11, hmap = ctx.saved_tensors
            dW = torch.empty(wshape, device=h.device) if (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) else None
            dcs = torch.empty(cshape, device=h.device) if dW is not None else None
            return None, dW, dcs, None, None, None          # no gradient for h from the duration branch: the fan-in sums two terms
        return patch(F._DrtDirect, bwd)

    def no_drt_bwd_data_only():
        def bwd(ctx, dD):
            B, Hm, Wm, C_, nsel, nheads, wshape, cshape = ctx.cfg
            h, W11, hmap = ctx.saved_tensors
            L = F.hip.lib()
            dW = torch.empty(wshape, dtype=torch.float32, device=h.device)
            dcs = torch.empty(cshape, dtype=torch.float32, device=h.device)
            ws = F.hip.workspace(L.sp_drt_direct_bwd_weight_workspace(B, Hm, Wm, C_, nsel), h.device, slot=0)
            rc = F.rows_ctx(ctx.step, B)
            F.check(L.sp_drt_direct_bwd_weight_rows(F.ptr(dD.contiguous()), F.ptr(h), F.ptr(hmap), B, Hm, Wm, C_, nsel, nheads, F.ptr(ws), F.ptr(dW),
                                                    F.ptr(dcs), F.ptr(rc.last) if rc is not None else None, int(ctx.step) if rc is not None else 0,
                                                    F.hip.stream()), "w")
            return None, dW, dcs, None, None, None
        return patch(F._DrtDirect, bwd)

    def no_sempool_bwd():
        def bwd(ctx, dout):
            amaps, vf, out = ctx.saved_tensors
            return torch.empty_like(amaps), None, None          # no gradient for vf from this push: its fan-in loses a term
        return patch(F._SemPool, bwd)

    def no_salgather_bwd():
        def bwd(ctx, dZ2):
            B, Hm, Wm, ldt, nsel, nsrc = ctx.cfg
            return torch.empty((B, Hm, Wm, ldt), dtype=torch.float32, device=dZ2.device), None, None, None, None
        return patch(F._SalGather, bwd)

    def no_listatt_bwd():
        def bwd(ctx, dmem):
            Lst, u, alpha = ctx.saved_tensors
            return torch.empty_like(Lst), (torch.empty_like(u) if ctx.needs_input_grad[1] else None)
        return patch(F._ListAtt, bwd)

    def no_rank1():
        old = F._lstm_rank1_backward

        def f(gates, c_prev, c, spcol, wc, dh, dc, need_dsp, need_dwc, *a, **k):
            dpre, dcp, _, _ = old(gates, c_prev, c, spcol, wc, dh, dc, False, False, *a, **k)
            return dpre, dcp, (torch.empty_like(spcol) if need_dsp else None), (torch.empty_like(wc) if need_dwc else None)
        F._lstm_rank1_backward = f
        return lambda: setattr(F, "_lstm_rank1_backward", old)

    def serial():
        F.ASYNC_DGRAD = False
        return lambda: setattr(F, "ASYNC_DGRAD", True)

    def several(*fs):
        def app():
            undos = [f() for f in fs]
            return lambda: [u() for u in undos]
        return app

    variants = [("base", None), ("no_drt_bwd", no_drt_bwd), ("no_drt_bwd_data_only", no_drt_bwd_data_only), ("no_sempool_bwd", no_sempool_bwd),
                ("no_salgather_bwd", no_salgather_bwd), ("no_listatt_bwd", no_listatt_bwd), ("no_rank1_grads", no_rank1),
                ("no_drt+sempool+salgather+listatt", several(no_drt_bwd, no_sempool_bwd, no_salgather_bwd, no_listatt_bwd)),
                ("serial_backward", serial), ("serial+no_drt_bwd", several(serial, no_drt_bwd))]
    res = {n: [] for n, _ in variants}
    timed()
    for _ in range(args.rounds):
        for name, app in variants:
            undo = app() if app is not None else None
            try:
                res[name].append(round(timed(), 2))
            finally:
                if undo is not None:
                    undo()
    base = sum(res["base"]) / len(res["base"])
    out = {n: {"ms": v, "delta_vs_base": round(sum(v) / len(v) - base, 2)} for n, v in res.items()}
    print(json.dumps({"tool": "chain_ablation (wrong results: timing only)", "steps": args.steps, "variants": out}))


if __name__ == "__main__":
    main()
