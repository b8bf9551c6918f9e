"""Diagnostic (GPU box): per-parameter gradient error of the HIP train step vs the fp64 oracle, next to the fp32 oracle's own
error, at the benchmark map size (320x512, 2 images, tame weights) -- for several settings of the fusion switches / GEMM back-ends.
    python tests/diagnostics/grad_error_table.py [T]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import oracle_state                      # noqa: E402
from oracle import scanpath_oracle as O               # noqa: E402
from scanpaths_amd import functional as F             # noqa: E402
from scanpaths_amd.models.loss import supervised_loss  # noqa: E402
from scanpaths_amd.models.scanpath_model import ScanpathModel  # noqa: E402
from scanpaths_amd.procedural import fill_module      # noqa: E402
from scanpaths_amd.spec import is_buffer              # noqa: E402
from scanpaths_amd.synth import make_batch            # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
Hm, Wm, NB, seed = 40, 64, 2, 21
DEV = "cuda:0"
torch.set_num_threads(min(64, os.cpu_count()))
b = make_batch("AiR", NB, 320, 512, T, seed=seed)
grads, losses = {}, {}
for dt, tag in ((torch.float64, "64"), (torch.float32, "32")):
    sd = oracle_state("AiR", "resnet50", seed, Hm, Wm, dtype=dt, family="tame")
    bd = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in b.items()}
    for k, v in sd.items():
        if v.is_floating_point() and not is_buffer(k):
            v.requires_grad_(True)
    tr = O.forward(sd, "AiR", bd["images"], bd["attention_maps"], bd["performances"], training=True, T=T)
    loss, _, _ = O.supervised_loss(tr, bd)
    loss.backward()
    grads[tag] = {k: v.grad.double() for k, v in sd.items() if v.requires_grad and v.grad is not None}
    losses[tag] = float(loss.detach())
    del sd, tr, loss
g64, g32 = grads["64"], grads["32"]
top = max(float(v.norm()) for v in g64.values())
print(f"T={T} loss64 {losses['64']:.6f} loss32 {losses['32']:.6f}  top grad norm {top:.3f}", flush=True)


def run(name, **sw):
    saved = {k: getattr(F, k) for k in sw}
    for k, v in sw.items():
        setattr(F, k, v)
    try:
        m = ScanpathModel("AiR", convLSTM_length=T, map_width=Wm, map_height=Hm, arch="resnet50")
        fill_module(m, seed=seed, family="tame")
        m = m.to(DEV).train()
        bd = {k: v.to(DEV) for k, v in b.items()}
        F.reset_fusion_counts()
        pred = m(bd["images"], bd["attention_maps"], bd["performances"])
        loss, _, _ = supervised_loss(pred, bd["scanpaths"], bd["durations"], bd["action_masks"], bd["duration_masks"], 1.0)
        loss.backward()
        torch.cuda.synchronize()
        rows = []
        for k, p in m.named_parameters():
            if k not in g64:
                continue
            got = p.grad.detach().cpu().double() if p.grad is not None else torch.zeros_like(g64[k])
            e, fl, n = float((got - g64[k]).norm()), float((g32[k] - g64[k]).norm()), float(g64[k].norm())
            rows.append((k, e, fl, n))
        big = [r for r in rows if r[3] > 1e-3 * top]
        worst = sorted(big, key=lambda r: -r[1] / max(r[2], 1e-30))[:int(os.environ.get('NROWS', '8'))]
        print(f"== {name}: loss err {abs(float(loss) - losses['64']):.2e} (oracle32 {abs(losses['32'] - losses['64']):.2e}); "
              f"counts {dict((k, v) for k, v in F.FUSION_COUNTS.items() if v)}")
        for k, e, fl, n in worst:
            print(f"   {k:45s} err/norm {e / n:.2e}  oracle32 err/norm {fl / n:.2e}  ratio {e / max(fl, 1e-30):6.1f}  norm {n:.3e}")
        for name_ in [d for d in os.environ.get("DETAIL", "").split(",") if d]:
            # is the error concentrated in a few output channels (the signature of single ReLU-mask flips) or spread out?
            p_ = dict(m.named_parameters())[name_]
            err = (p_.grad.detach().cpu().double() - g64[name_])
            e32 = (g32[name_] - g64[name_])
            ch = err.flatten(1).pow(2).sum(1) if err.dim() > 1 else err.pow(2)
            ch32 = e32.flatten(1).pow(2).sum(1) if e32.dim() > 1 else e32.pow(2)
            tk = torch.topk(ch, min(5, ch.numel()))
            print(f"   DETAIL {name_}: share of squared error in the top-5 output channels {[round(float(v / ch.sum()), 4) for v in tk.values]} "
                  f"(channels {tk.indices.tolist()}); oracle32: {[round(float(v / ch32.sum()), 4) for v in torch.topk(ch32, min(5, ch32.numel())).values]}; "
                  f"err norm without the top-2 channels {float((ch.sum() - tk.values[:2].sum()).clamp(min=0).sqrt()):.3e} vs oracle32 {float(ch32.sum().sqrt()):.3e}")
        sys.stdout.flush()
        return {k: (e, fl, n) for k, e, fl, n in rows}
    finally:
        for k, v in saved.items():
            setattr(F, k, v)


CONFIGS = {
    "default_bs2_costmodel": ("default (cost model at bs 2)", {}),
    "bench_path": ("bench path (COST_M_SCALE 16)", dict(COST_M_SCALE=16.0)),
    "no_lstm_bwd_split": ("bench path, LSTM_BWD_SPLIT off", dict(COST_M_SCALE=16.0, LSTM_BWD_SPLIT=False)),
    "no_bn_split": ("bench path, BN_SPLIT off", dict(COST_M_SCALE=16.0, BN_SPLIT=False)),
    "no_fuse": ("bench path, FUSE_GATE_LSTM off", dict(COST_M_SCALE=16.0, FUSE_GATE_LSTM=False)),
    "no_hplanes": ("bench path, LSTM_H_PLANES off", dict(COST_M_SCALE=16.0, LSTM_H_PLANES=False)),
    "no_amax_hint": ("bench path, FUSED_AMAX off", dict(COST_M_SCALE=16.0, FUSED_AMAX=False)),
    "no_defer": ("bench path, DEFER_WGRAD off", dict(COST_M_SCALE=16.0, DEFER_WGRAD=False)),
    "no_channel_scales": ("bench path, CHANNEL_SCALES off", dict(COST_M_SCALE=16.0, CHANNEL_SCALES=False)),
    "no_rank1_split": ("bench path, RANK1_DSP_SPLIT / RANK1_DWC_SPLIT off", dict(COST_M_SCALE=16.0, RANK1_DSP_SPLIT=False, RANK1_DWC_SPLIT=False)),
    "no_skip_dpre": ("bench path, LSTM_SKIP_DPRE off", dict(COST_M_SCALE=16.0, LSTM_SKIP_DPRE=False)),
    "bf16x3": ("bf16x3 back-end", dict(COST_M_SCALE=16.0, SPLIT_SCHEME="bf16x3")),
    "fp32": ("fp32 MFMA back-end", dict(USE_BF16X3=False)),
}
out = {}
for key in (os.environ.get("CONFIGS", ",".join(CONFIGS)).split(",")):
    out[key] = run(*[CONFIGS[key][0]], **CONFIGS[key][1])
os.makedirs(os.path.join(ROOT, "gpurun_out", "diag"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "diag", f"grad_error_table_T{T}.json"), "w"))
