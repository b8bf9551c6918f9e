"""GPU diagnostic: layer-wise error of the HIP encoder/decoder vs the fp64 oracle (and the fp32 oracle's own error)."""
import sys, os
import numpy as np
import torch
import torch.nn.functional as TF
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from helpers import case_inputs, load_golden, max_err, oracle_state
from test_model_gpu import _build, DEV
from oracle import scanpath_oracle as O
from scanpaths_amd import functional as F

name = sys.argv[1] if len(sys.argv) > 1 else "air_eval_T4"
from scanpaths_amd import config as _sp_config
_sp_config.honour_env_for_tools()      # SP_SPLIT_SCHEME / SP_NO_SPLIT of the command line, through the switchboard's checks
training = "train" in name
meta, g = load_golden(name)
b = case_inputs(meta, torch.float32)
model = _build(meta); model.train(training)
sd64 = oracle_state(meta["task"], meta["arch"], meta["weight_seed"])
sd32 = {k: (v.float() if v.is_floating_point() else v) for k, v in sd64.items()}

def rep(tag, hip_nhwc, r64, r32):
    h = hip_nhwc.detach().cpu().double()
    if h.dim() == 4: h = h.permute(0, 3, 1, 2)
    sc = r64.abs().max().item()
    print(f"{tag:28s} scale {sc:10.3e} rms {r64.pow(2).mean().sqrt().item():9.3e}  hip {((h - r64).abs().max().item())/sc:.2e}  oracle32 {((r32.double() - r64).abs().max().item())/sc:.2e}")

# ---- encoder, stage by stage (mirrors oracle.encoder) ----
def enc_stages(sd, x, dt):
    outs = []
    pfx = "resnet."
    x = TF.conv2d(x, sd[pfx + "0.weight"], None, stride=2, padding=3); outs.append(("stem_conv", x))
    x = TF.relu(O._bn(sd, pfx + "1", x, training, None)); outs.append(("stem_bn", x))
    x = TF.max_pool2d(x, 3, 2, 0, ceil_mode=True); outs.append(("maxpool", x))
    return outs

with torch.no_grad():
    img = b["images"]
    o64 = enc_stages(sd64, img.double(), torch.float64)
    o32 = enc_stages(sd32, img, torch.float32)
    r = model.resnet
    x = F.nchw_to_nhwc(img.to(DEV), 4)
    w0 = F.pad_last(r[0].weight.permute(0, 2, 3, 1), 4).permute(0, 3, 1, 2)
    x = F.conv2d(x, w0, None, stride=2, pad=3); rep("stem_conv", x, o64[0][1], o32[0][1])
    x = model._bn(r[1], x); rep("stem_bn", x, o64[1][1], o32[1][1])
    x = F.maxpool3s2(x); rep("maxpool", x, o64[2][1], o32[2][1])
    e64 = O.encoder(sd64, img.double(), meta["arch"], training)
    e32 = O.encoder(sd32, img, meta["arch"], training)
    model.train(training)
    eh = model.encode(img.to(DEV)); rep("encoder_out", eh, e64, e32)
    v64 = TF.relu(O._conv(sd64, "sal_conv", e64, padding=1)); v32 = TF.relu(O._conv(sd32, "sal_conv", e32, padding=1))
    vh = F.conv2d(eh, model.sal_conv.weight, model.sal_conv.bias, pad=1, relu=True); rep("vf", vh, v64, v32)
    # vf from the SAME (oracle fp64 -> fp32) encoder output: isolates sal_conv
    vh2 = F.conv2d(e64.float().permute(0, 2, 3, 1).contiguous().to(DEV), model.sal_conv.weight, model.sal_conv.bias, pad=1, relu=True)
    v32b = TF.relu(O._conv(sd32, "sal_conv", e64.float(), padding=1))
    rep("vf | exact enc input", vh2, v64, v32b)
