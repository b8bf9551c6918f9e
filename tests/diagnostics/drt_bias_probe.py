"""Diagnostic (GPU box): is the gradient error of object_head.drt_layer_1.bias (a SCALAR: the sum of the duration-branch gradients over
all sites, samples, heads and steps) on the golden case air_tame_train_T16 a ReLU-kink event of the duration branch?  For each GEMM
back-end: the error of that gradient against the reference's fp64 value, the duration pre-activations closest to zero, and the sites
whose ReLU mask differs from the fp32-MFMA run.     python tests/diagnostics/drt_bias_probe.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import case_inputs, load_golden          # noqa: E402
from scanpaths_amd import functional as F             # noqa: E402
from scanpaths_amd.models.loss import supervised_loss  # noqa: E402
from scanpaths_amd.models.scanpath_model import ScanpathModel  # noqa: E402
from scanpaths_amd.procedural import fill_module      # noqa: E402

DEV = "cuda:0"
meta, g = load_golden("air_tame_train_T16")
b = case_inputs(meta, torch.float32)
names = meta["param_names"]
key = "ref64/grad/object_head.drt_layer_1.bias"
ref = float(np.asarray(g[key]).reshape(-1)[0]) if key in g else None
r32 = float(np.asarray(g[key.replace("ref64", "ref32")]).reshape(-1)[0]) if key in g else None
print("reference fp64", ref, "fp32", r32)


def run(**sw):
    saved = {k: getattr(F, k) for k in sw}
    for k, v in sw.items():
        setattr(F, k, v)
    cap = []
    orig = F.head_finish

    def hf(Z2, cb, w2, b2, nh, HC, softmax, per_sample=False, dpre=None):
        cap.append((dpre.detach().double().cpu(), cb.detach().double().cpu()))
        return orig(Z2, cb, w2, b2, nh, HC, softmax, per_sample=per_sample, dpre=dpre)
    F.head_finish = hf
    try:
        m = ScanpathModel(meta["task"], convLSTM_length=meta["T"], map_width=40, map_height=30, arch=meta["arch"])
        fill_module(m, seed=meta["weight_seed"], family=meta.get("weight_family", "default"))
        m = m.to(DEV).train()
        pred = m(b["images"].to(DEV), b["attention_maps"].to(DEV), b["performances"].to(DEV))
        loss, _, _ = supervised_loss(pred, b["scanpaths"].to(DEV), b["durations"].to(DEV), b["action_masks"].to(DEV),
                                     b["duration_masks"].to(DEV), 1.0)
        loss.backward()
        torch.cuda.synchronize()
        gb = float(dict(m.named_parameters())["object_head.drt_layer_1.bias"].grad.double().cpu().reshape(-1)[0])
        pre = torch.stack([d + c[:, 2 + 49].view(-1, 1, 1) for d, c in cap], 0)          # [T, nh, B, S] duration pre-activation incl. bias
        return gb, pre
    finally:
        F.head_finish = orig
        for k, v in saved.items():
            setattr(F, k, v)


gb0, pre0 = run(USE_BF16X3=False)
print(f"fp32-MFMA  : grad {gb0:.9e}  err vs fp64 {abs(gb0 - ref):.2e}   min |pre-activation| {float(pre0.abs().min()):.2e}")
for name, sw in (("f16x2", {}), ("bf16x3", dict(SPLIT_SCHEME="bf16x3"))):
    gb, pre = run(**sw)
    flips = ((pre > 0) != (pre0 > 0)).nonzero()
    print(f"{name:10s} : grad {gb:.9e}  err vs fp64 {abs(gb - ref):.2e}   min |pre-activation| {float(pre.abs().min()):.2e}   mask flips vs the fp32-MFMA run "
          f"{flips.tolist()}  max |pre - pre_fp32| {float((pre - pre0).abs().max()):.2e}")
    for f in flips.tolist():
        print("      flipped site", f, "pre", float(pre[tuple(f)]), "fp32-run pre", float(pre0[tuple(f)]))
