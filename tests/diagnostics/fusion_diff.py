"""Which round-2 fusion moves the training-step gradients, and by how much compared with the sensitivity of the step to rounding
(same step on the 3xbf16 back-end)?   python3 tests/diagnostics/fusion_diff.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from scanpaths_amd import functional as F  # noqa: E402
from scanpaths_amd.models.baseline_attention import baseline  # noqa: E402
from scanpaths_amd.models.loss import supervised_loss  # noqa: E402
from scanpaths_amd.procedural import fill_module  # noqa: E402
from scanpaths_amd.synth import make_batch  # noqa: E402

DEV = "cuda:0"
T = 3
FAMILY = os.environ.get("FAMILY", "tame")
b = {k: v.to(DEV) for k, v in make_batch("AiR", 4, 256, 512, T, seed=5).items()}
FLAGS = ("BN_SPLIT", "GRAD_MERGE", "FUSE_GATE_LSTM")


def run(on=(), scheme=None):
    for f in FLAGS:
        setattr(F, f, f in on)
    old = F.SPLIT_SCHEME
    if scheme:
        F.SPLIT_SCHEME = scheme
    m = baseline(convLSTM_length=T, map_width=64, map_height=32)
    fill_module(m, 5, family=FAMILY)
    m = m.to(DEV).train()
    pred = m(b["images"], b["attention_maps"], b["performances"])
    loss, _, _ = supervised_loss(pred, b["scanpaths"], b["durations"], b["action_masks"], b["duration_masks"], 1.0)
    loss.backward()
    F.SPLIT_SCHEME = old
    return float(loss), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}


def diff(a, c):
    gmax = max(float(v.abs().max()) for v in c.values())
    rows = sorted(((float((a[k] - c[k]).abs().max()) / max(float(c[k].abs().max()), 1e-3 * gmax), k) for k in c), reverse=True)
    return rows[:4], sum(r[0] for r in rows) / len(rows)


l0, g0 = run(())
out = {"loss_off": l0}
for name, kw in (("off_again", dict(on=())), ("bf16x3_backend", dict(on=(), scheme="bf16x3")), ("BN_SPLIT", dict(on=("BN_SPLIT",))),
                 ("GRAD_MERGE", dict(on=("GRAD_MERGE",))), ("FUSE_GATE_LSTM", dict(on=("FUSE_GATE_LSTM",))), ("all", dict(on=FLAGS))):
    l, g = run(**kw)
    worst, mean = diff(g, g0)
    out[name] = {"loss": l, "worst": worst, "mean_rel": mean}
    print(name, l, worst, mean, flush=True)
print(json.dumps(out))
