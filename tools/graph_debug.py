"""Diagnose HIP-graph replay vs eager at different input amplitudes (encoder only / full model; SP_NO_AMAX_HINT, back-ends)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from scanpaths_amd import functional as F
from scanpaths_amd.models.baseline_attention import baseline
from scanpaths_amd.procedural import fill_module
from scanpaths_amd.synth import make_batch

DEV = "cuda:0"
T = 2
m = baseline(convLSTM_length=T)
fill_module(m, 2)
m = m.to(DEV).eval()
b = make_batch("AiR", 2, 240, 320, T, seed=2)
img, att = b["images"].to(DEV), b["attention_maps"].to(DEV)
base = make_batch("AiR", 2, 240, 320, T, seed=7)["images"].to(DEV)


def capture(fn):
    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fn()
    return g, out


def stages(x):
    """encoder prefix outputs for localisation"""
    r = m.resnet
    outs = {}
    x0 = F.nchw_to_nhwc(x, 4)
    w0 = F.pad_last(r[0].weight.permute(0, 2, 3, 1), 4).permute(0, 3, 1, 2)
    c1 = F.conv2d(x0, w0, None, stride=2, pad=3)
    outs["conv1"] = c1
    b1 = m._bn(r[1], c1)
    outs["bn1"] = b1
    p = F.maxpool3s2(b1)
    outs["pool"] = p
    blk = r[4][0]
    o = F.conv2d(p, blk.conv1.weight, None)
    outs["l1c1"] = o
    o = m._bn(blk.bn1, o)
    outs["l1b1"] = o
    o = F.conv2d(o, blk.conv2.weight, None, pad=1)
    outs["l1c2"] = o
    return outs


for name, fn in (("stages", lambda: stages(img)), ("encode", lambda: {"enc": m.encode(img)}), ("model", lambda: m(img, att))):
    img.copy_(b["images"].to(DEV))
    g, out = capture(fn)
    for amp in (1.0, 100.0, 0.01, 1.0):
        img.copy_(base * amp)
        g.replay()
        torch.cuda.synchronize()
        got = {k: v.clone() for k, v in out.items()}
        with torch.no_grad():
            ref = fn()
        torch.cuda.synchronize()
        line = []
        for k in ref:
            d = float((got[k] - ref[k]).abs().max())
            s = float(ref[k].abs().max())
            line.append(f"{k}: {'==' if torch.equal(got[k], ref[k]) else f'diff {d:.2e} (scale {s:.2e})'}")
        print(f"[{name}] amp {amp}: " + "; ".join(line), flush=True)
