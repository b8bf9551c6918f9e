"""GPU diagnostic (not a test): per-key / per-parameter error report of the HIP model against the goldens."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from helpers import case_inputs, load_golden, max_err
from test_model_gpu import _build, _call, DEV
from scanpaths_amd.models.loss import supervised_loss
from scanpaths_amd import functional as F
from scanpaths_amd import config as _sp_config
_sp_config.honour_env_for_tools()      # SP_SPLIT_SCHEME / SP_NO_SPLIT of the command line, through the switchboard's checks

def fwd(name):
    meta, g = load_golden(name)
    b = case_inputs(meta, torch.float32)
    model = _build(meta)
    model.train(meta["mode"] == "train")
    with torch.set_grad_enabled(meta["mode"] == "train"):
        pred = _call(model, meta, b)
    for k, v in pred.items():
        ref = g["ref64/" + k]
        fl = max_err(g["ref32/" + k], ref)
        e = max_err(v, ref)
        extra = ""
        if v.dim() >= 2 and v.shape[1] == meta["T"]:
            per_t = [max_err(v[:, t], ref[:, t]) for t in range(meta["T"])]
            per_t32 = [max_err(g["ref32/" + k][:, t], ref[:, t]) for t in range(meta["T"])]
            extra = "  per-step hip " + " ".join(f"{x:.1e}" for x in per_t) + " | ref32 " + " ".join(f"{x:.1e}" for x in per_t32)
        print(f"{name}:{k}: hip {e:.2e} ref32 {fl:.2e} scale {np.abs(ref).max():.2f}{extra}")
    return meta, g, b, model, pred

def train(name):
    meta, g, b, model, pred = fwd(name)
    loss, la, ld = supervised_loss(pred, b["scanpaths"].to(DEV), b["durations"].to(DEV), b["action_masks"].to(DEV),
                                   b["duration_masks"].to(DEV), 1.0)
    print("loss", loss.item(), la.item(), ld.item(), "ref64", g["ref64/loss"], "ref32", g["ref32/loss"])
    loss.backward()
    names = meta["param_names"]
    params = dict(model.named_parameters())
    gn = np.array([params[k].grad.norm().item() if params[k].grad is not None else 0.0 for k in names])
    gref, g32 = g["ref64/grad_norms"], g["ref32/grad_norms"]
    order = np.argsort(-np.abs(gn - gref))[:25]
    for i in order:
        print(f"  {names[i]:45s} hip {gn[i]:.6e} ref64 {gref[i]:.6e} ref32 {g32[i]:.6e}  rel {abs(gn[i]-gref[i])/(gref[i]+1e-30):.2e} (ref32 rel {abs(g32[i]-gref[i])/(gref[i]+1e-30):.2e})")
    worst = []
    for k in g:
        if k.startswith("ref64/grad/") or k.startswith("ref64/gradsample/"):
            full = k.startswith("ref64/grad/")
            pname = k.split("/", 2)[2]
            gg = params[pname].grad
            if gg is None:
                gg = torch.zeros_like(params[pname])
            if not full:
                gg = gg.flatten()[::max(1, gg.numel() // 512)][:512]
            ref = g[k]
            worst.append((max_err(gg, ref) / (np.abs(ref).max() + 1e-30), max_err(g[k.replace("ref64", "ref32")], ref) / (np.abs(ref).max() + 1e-30), pname))
    worst.sort(reverse=True)
    for w in worst[:25]:
        print(f"  elementwise rel err {w[0]:.2e} (ref32 {w[1]:.2e}) {w[2]}")

if __name__ == "__main__":
    for n in sys.argv[1:]:
        (train if "train" in n else fwd)(n)
