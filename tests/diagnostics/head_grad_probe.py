"""Diagnostic (GPU box): where does the gradient error of performance_sal_layer.True.weight on the 2xfp16 back-end come from?
The bench path (320x512, 2 images, tame weights, T steps) on the fp32-MFMA back-end (reference) and on the 2xfp16 back-end: the
gradients of the composed head filters G (per row: terminate map, action map, 49 duration taps), of the tap GEMM weight Wsal and of
the duration-window weights W11 are captured and compared row by row.     python tests/diagnostics/head_grad_probe.py [T]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from scanpaths_amd import functional as F             # noqa: E402
from scanpaths_amd.models import scanpath_model as SM  # noqa: E402
from scanpaths_amd.models.loss import supervised_loss  # noqa: E402
from scanpaths_amd.procedural import fill_module      # noqa: E402
from scanpaths_amd.synth import make_batch            # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 16
Hm, Wm, NB, seed = 40, 64, 2, 21
DEV = "cuda:0"
b = {k: v.to(DEV) for k, v in make_batch("AiR", NB, 320, 512, T, seed=seed).items()}
print("performances", b["performances"].tolist())


def run(**sw):
    saved = {k: getattr(F, k) for k in sw}
    for k, v in sw.items():
        setattr(F, k, v)
    cap = {}
    orig_compose, orig_c11 = SM.ScanpathModel._compose_heads, F.compose11

    def compose(self, convs):
        G, cb = orig_compose(self, convs)
        G.retain_grad()
        cap["G"] = G
        return G, cb

    def c11(G, cb, nh, HC, hw):
        W11, cbsum = orig_c11(G, cb, nh, HC, hw)
        W11.retain_grad()
        cap["W11"] = W11
        return W11, cbsum
    orig_sg_bwd = F._SalGather.backward
    cap["dT"] = []

    def sg_bwd(ctx, dZ2):
        outs = orig_sg_bwd(ctx, dZ2)
        cap["dT"].append((dZ2.detach().double().cpu(), outs[0].detach().double().cpu()))
        return outs
    F._SalGather.backward = staticmethod(sg_bwd)
    SM.ScanpathModel._compose_heads, F.compose11 = compose, c11
    try:
        m = SM.ScanpathModel("AiR", convLSTM_length=T, map_width=Wm, map_height=Hm, arch="resnet50")
        fill_module(m, seed=seed, family="tame")
        m = m.to(DEV).train()
        pred = m(b["images"], b["attention_maps"], b["performances"])
        loss, _, _ = supervised_loss(pred, b["scanpaths"], b["durations"], b["action_masks"], b["duration_masks"], 1.0)
        loss.backward()
        torch.cuda.synchronize()
        out = {"dG": cap["G"].grad.detach().double().cpu(), "dW11": cap["W11"].grad.detach().double().cpu(), "dT": cap["dT"]}
        for k, p in m.named_parameters():
            if k.startswith("performance_sal_layer") or k.startswith("object_head") or k in ("lstm.input_x.bias", "lstm.memory_h.bias"):
                out[k] = p.grad.detach().double().cpu()
        return out
    finally:
        F._SalGather.backward = staticmethod(orig_sg_bwd)
        SM.ScanpathModel._compose_heads, F.compose11 = orig_compose, orig_c11
        for k, v in saved.items():
            setattr(F, k, v)


ref = run(USE_BF16X3=False)
_pays = F._b3_pays


def _no_hgate_dgrad(M, N, K, Kc, nbatch=1, **kw):      # the h-gate conv's data gradient (N = 512, K = 9 * 2048) on the fp32-MFMA kernel
    return False if (N == 512 and K == 18432) else _pays(M, N, K, Kc, nbatch, **kw)


def _no_hgate_fwd(M, N, K, Kc, nbatch=1, **kw):        # ... and its forward (N = 2048, K = 4608) as well
    return False if ((N == 512 and K == 18432) or (N == 2048 and K == 4608)) else _pays(M, N, K, Kc, nbatch, **kw)


for name, sw in (("f16x2 bench path", dict(COST_M_SCALE=16.0)), ("f16x2, CHANNEL_SCALES off", dict(COST_M_SCALE=16.0, CHANNEL_SCALES=False)),
                 ("f16x2, rank-1 gradients on fp32", dict(COST_M_SCALE=16.0, RANK1_DSP_SPLIT=False, RANK1_DWC_SPLIT=False)),
                 ("f16x2, h-gate data gradient on fp32 MFMA", dict(COST_M_SCALE=16.0, _b3_pays=_no_hgate_dgrad)),
                 ("f16x2, h-gate forward + data gradient on fp32 MFMA", dict(COST_M_SCALE=16.0, _b3_pays=_no_hgate_fwd, FUSE_GATE_LSTM=False)),
                 ("bf16x3", dict(COST_M_SCALE=16.0, SPLIT_SCHEME="bf16x3"))):
    got = run(**sw)
    print("==", name)
    for k in got:
        if k in ("dG", "dW11", "dT"):
            continue
        d = (got[k] - ref[k]).norm() / ref[k].norm().clamp_min(1e-300)
        print(f"   {k:45s} rel diff to the fp32-MFMA run {float(d):.2e}  norm {float(ref[k].norm()):.3e}")
    dG, rG = got["dG"], ref["dG"]                     # [nh*HC, 512, 5, 5]
    HC = dG.shape[0] // 2
    for hd, hn in ((0, "True/good"), (1, "False/poor")):
        for r, rn in ((0, "terminate map"), (1, "action map")):
            a, c = dG[hd * HC + r], rG[hd * HC + r]
            print(f"   dG {hn:10s} row {rn:14s}: rel diff {float((a - c).norm() / c.norm().clamp_min(1e-300)):.2e}  norm {float(c.norm()):.3e}")
        a, c = dG[hd * HC + 2: hd * HC + 51], rG[hd * HC + 2: hd * HC + 51]
        print(f"   dG {hn:10s} 49 duration-tap rows   : rel diff {float((a - c).norm() / c.norm().clamp_min(1e-300)):.2e}  norm {float(c.norm()):.3e}")
    # per backward call of the tap gather (last decode step first): dZ2 [B,Hm,Wm,4] = (True term, True act, False term, False act) and dT
    for i, ((z, t), (zr, tr)) in enumerate(zip(got["dT"], ref["dT"])):
        rel = lambda a_, c_: float((a_ - c_).norm() / c_.norm().clamp_min(1e-300))
        zt, ztr = z[..., 1], zr[..., 1]
        flips = int(((zt != 0) != (ztr != 0)).sum())
        print(f"   gather bwd call {i:2d}: dZ2 True-act rel diff {rel(zt, ztr):.2e} (norm {float(ztr.norm()):.2e}, nonzero {int((ztr != 0).sum())}, mask flips {flips}); "
              f"False-act {rel(z[..., 3], zr[..., 3]):.2e} (norm {float(zr[..., 3].norm()):.2e}); dT True-act cols {rel(t[..., 25:50], tr[..., 25:50]):.2e}")
    a_t, c_t = dG[1], rG[1]
    e_ = (a_t - c_t)
    print(f"   dG True action row: share of squared error in the 5 worst input channels {[round(float(v), 3) for v in (e_.pow(2).sum((1, 2)).topk(5).values / e_.pow(2).sum())]}; "
          f"per 5x5 tap {[round(float(v), 3) for v in (e_.pow(2).sum(0).flatten() / e_.pow(2).sum())]}")
    a, c = got["dW11"], ref["dW11"]
    for hd in range(a.shape[0]):
        print(f"   dW11 head {hd}: rel diff {float((a[hd] - c[hd]).norm() / c[hd].norm().clamp_min(1e-300)):.2e}  norm {float(c[hd].norm()):.3e}")
    sys.stdout.flush()
