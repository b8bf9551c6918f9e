#!/usr/bin/env python3
"""debug: where the batched duration branch leaves the per-step form (bitwise)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from scanpaths_amd import functional as F
from scanpaths_amd.models.scanpath_model import ScanpathModel, HC
from scanpaths_amd.procedural import fill_module
from scanpaths_amd.synth import make_batch
DEV = "cuda:0"
T, NB = 6, 5
task = sys.argv[1] if len(sys.argv) > 1 else "AiR"
m = ScanpathModel(task, convLSTM_length=T, map_width=64, map_height=40, arch="resnet18")
fill_module(m, seed=4, family="tame")
m = m.to(DEV).train()
b = {k: v.to(DEV) for k, v in make_batch(task, NB, 320, 512, T, seed=4).items()}
batches = []
orig = F.DrtBatch
class Rec(orig):
    def __init__(self, T):
        super().__init__(T); batches.append(self)
F.DrtBatch = Rec
F.COST_M_SCALE = 32.0 / 5
with torch.no_grad():
    args = (b["images"], b["attention_maps"], b["performances"]) if task == "AiR" else (b["images"], b["attention_maps"], b["tasks"])
    F.DRT_BATCHED = True
    pa = m(*args)
    db = batches[-1]
    F.DRT_BATCHED = False
    pb = m(*args)
    for k in pa:
        d = (pa[k] - pb[k]).abs().max().item()
        print(k, "max diff", d, "n differing", int((pa[k] != pb[k]).sum()))
    # sites per step from the recorded operands, one launch each, vs the batch's buffer
    for t, (h, W11, cbsum, hmap, nsel) in enumerate(db.items):
        D1 = F.drt_direct(h, W11, cbsum, hmap, nsel)
        print("step", t, "Dpre diff", (D1 - db.D[t]).abs().max().item(), "n", int((D1 != db.D[t]).sum()), "h contiguous in buf",
              h.data_ptr() == db.buf[t].data_ptr())

# ---- timing of the one-launch site kernel at the benchmark size (T x B = 512 hidden states of 40 x 64 x 512) --------------------------------
if len(sys.argv) > 2 and sys.argv[2] == "time":
    from scanpaths_amd import hip
    L = hip.lib()
    Tn, B, Hm, Wm, C = 16, 32, 40, 64, 512
    h = torch.randn(Tn * B, Hm, Wm, C, device=DEV)
    ncls = L.sp_head_num_classes(Hm, Wm)
    W11 = torch.randn(2, ncls, 121, C, device=DEV) * 0.01
    cbsum = torch.zeros(2, ncls, device=DEV)
    hmap = torch.arange(2, dtype=torch.int32, device=DEV).repeat(Tn * B, 1).contiguous()
    S = 8 * 13
    D = torch.empty(2, Tn * B, S, device=DEV)
    for nb in (Tn * B, B):
        for _ in range(2):
            hip.check(L.sp_drt_direct_fwd(hip.ptr(h), hip.ptr(W11), hip.ptr(cbsum), hip.ptr(hmap), nb, Hm, Wm, C, 2, hip.ptr(D), hip.stream()), "f")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            hip.check(L.sp_drt_direct_fwd(hip.ptr(h), hip.ptr(W11), hip.ptr(cbsum), hip.ptr(hmap), nb, Hm, Wm, C, 2, hip.ptr(D), hip.stream()), "f")
        e1.record(); torch.cuda.synchronize()
        print(f"sp_drt_direct_fwd rows {nb}: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us per launch")
