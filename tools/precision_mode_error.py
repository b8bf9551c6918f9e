"""Measured parity error of a GEMM precision mode on the tame goldens (tests/golden/*_tame_*): per decode step
err(HIP, reference fp64) / scale for the fp32-faithful default and for the throughput mode (SP_SPLIT_SCHEME=f16x1).
    python3 tools/precision_mode_error.py            -> JSON on stdout (run once per mode; the mode is fixed at import)
    SP_SPLIT_SCHEME=f16x1 python3 tools/precision_mode_error.py"""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import case_inputs, load_golden  # noqa: E402
from scanpaths_amd import config as _sp_config  # noqa: E402
_sp_config.honour_env_for_tools()      # the mode named on the command line (SP_SPLIT_SCHEME=f16x1), through the switchboard's checks
from scanpaths_amd import functional as F  # noqa: E402
from scanpaths_amd.models.scanpath_model import ScanpathModel  # noqa: E402
from scanpaths_amd.procedural import fill_module  # noqa: E402

out = {"mode": "f16x1 (throughput)" if F.THROUGHPUT_MODE else F.SPLIT_SCHEME + " (fp32-faithful)", "cases": {}}
for name in ("air_tame_eval_T16", "air_tame_train_T16", "coco_tame_eval_T6", "osie_r18_tame_eval_T8"):
    meta, g = load_golden(name)
    b = case_inputs(meta, torch.float32)
    m = ScanpathModel(meta["task"], convLSTM_length=meta["T"], arch=meta["arch"])
    fill_module(m, seed=meta["weight_seed"], family="tame")
    m = m.cuda()
    m.train(meta["mode"] == "train")
    img = b["images"].cuda()
    with torch.no_grad():
        if meta["task"] == "AiR":
            pred = m(img, b["attention_maps"].cuda(), b["performances"].cuda() if m.training else None)
        elif meta["task"] == "OSIE":
            pred = m(img)
        else:
            pred = m(img, b["attention_maps"].cuda(), b["tasks"].cuda())
    res = {}
    for k, v in pred.items():
        ref, r32 = torch.as_tensor(g["ref64/" + k]), torch.as_tensor(g["ref32/" + k])
        scale = float(ref.abs().max())
        v = v.detach().cpu().double()
        T = meta["T"]
        err = [float((v[:, t] - ref[:, t]).abs().max()) / scale for t in range(T)]
        floor = [float((r32[:, t].double() - ref[:, t]).abs().max()) / scale for t in range(T)]
        am = None
        if k.endswith("all_actions_prob") or k == "actions":
            am = int((v.argmax(-1) == ref.argmax(-1)).sum()), int(ref.shape[0] * T)
        res[k] = {"scale": scale, "max_err_over_scale": max(err), "ref32_max_noise_over_scale": max(floor),
                  "err_over_scale_per_step": [round(e, 9) for e in err], "argmax_equal": am}
    out["cases"][name] = res
print(json.dumps(out))
