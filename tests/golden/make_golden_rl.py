"""Golden vectors for the RL-reward glue from the REAL reference, build container only:

    python tests/golden/make_golden_rl.py

Runs /root/reference/AiR/utils/evaluation.py::pairs_eval_scanmatch_performance_related (:361-422) and
gtpairs_eval_scanmatch_performance_related (:425-576) and /root/reference/AiR/models/loss.py::LogAction / LogDuration
(:34-45) on seeded random inputs.  multimatch_gaze / tqdm / cv2 / matplotlib are imported by those modules but not used by
these functions; they are stubbed.  Writes tests/golden/rl.npz."""
import os
import sys
import types

import numpy as np
import torch

for name in ("multimatch_gaze", "tqdm", "cv2", "matplotlib", "matplotlib.pyplot"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["tqdm"].tqdm = lambda x, **k: x
sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
sys.path.insert(0, "/root/reference/AiR")
from utils.evaltools.scanmatch import ScanMatch                                        # noqa: E402
from utils.evaluation import (gtpairs_eval_scanmatch_performance_related,               # noqa: E402
                              pairs_eval_scanmatch_performance_related)
from models.loss import LogAction, LogDuration                                         # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
DT = {"names": ("start_x", "start_y", "duration"), "formats": ("f8", "f8", "f8")}


def scanpath(g, empty_wd=False):
    L = int(g.integers(1, 12))
    fv = np.zeros(L, dtype=DT)
    fv["start_x"], fv["start_y"] = g.uniform(0, 320, L), g.uniform(0, 240, L)
    fv["duration"] = g.uniform(0.0, 0.02, L) if empty_wd else g.uniform(0.05, 0.9, L)      # seconds
    return fv


def main():
    g = np.random.Generator(np.random.PCG64(77))
    N = 6
    gt, perf = [], []
    for i in range(N):
        n = [4, 3, 1, 5, 2, 2][i]
        gt.append([scanpath(g, empty_wd=(i == 4 and k == 0)) for k in range(n)])
        perf.append([bool(v) for v in ([True, False, True, False], [True, True, True], [False], [True, True, False, False, True],
                                       [True, False], [False, False])[i]])
    pred = [scanpath(g) for _ in range(N)]
    wd = ScanMatch(Xres=320, Yres=240, Xbin=16, Ybin=12, Offset=(0, 0), TempBin=50, Threshold=3.5)
    wod = ScanMatch(Xres=320, Yres=240, Xbin=16, Ybin=12, Offset=(0, 0), Threshold=3.5)
    out = {}
    flat = [fv for lst in gt for fv in lst]
    out["gt_fix"] = np.concatenate([np.stack([f["start_x"], f["start_y"], f["duration"]], 1) for f in flat], 0)
    out["gt_len"] = np.array([len(f) for f in flat])
    out["gt_count"] = np.array([len(l) for l in gt])
    out["perf"] = np.array([int(p) for l in perf for p in l])
    out["pred_fix"] = np.concatenate([np.stack([f["start_x"], f["start_y"], f["duration"]], 1) for f in pred], 0)
    out["pred_len"] = np.array([len(f) for f in pred])
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for given in (True, False):
            same, diff, acc = pairs_eval_scanmatch_performance_related(gt, pred, wd, wod, perf, given)
            tag = "good" if given else "poor"
            out[f"pairs_{tag}_same"], out[f"pairs_{tag}_diff"], out[f"pairs_{tag}_accept"] = same, diff, np.array(int(acc))
        good, poor, gvp = gtpairs_eval_scanmatch_performance_related(gt, wd, wod, perf)
    out["gtpairs_good"], out["gtpairs_poor"], out["gtpairs_diff"] = good, poor, gvp
    # ---- LogAction / LogDuration with gradients ----
    tg = torch.Generator().manual_seed(5)
    B, T = 5, 7
    p = (torch.rand(B, T, generator=tg, dtype=torch.float64) * 0.9 + 1e-3).requires_grad_(True)
    amask = (torch.rand(B, T, generator=tg) < 0.7).double()
    d = torch.rand(B, T, generator=tg, dtype=torch.float64) * 0.8 + 0.05
    mu = torch.randn(B, T, generator=tg, dtype=torch.float64).requires_grad_(True)
    s2 = (torch.rand(B, T, generator=tg, dtype=torch.float64) + 0.2).requires_grad_(True)
    dmask = (torch.rand(B, T, generator=tg) < 0.6).double()
    w = torch.randn(B, generator=tg, dtype=torch.float64)
    la = LogAction(p, amask)
    ld = LogDuration(d, mu, s2, dmask)
    ((la * w).sum() + (ld * w).sum() * 0.5).backward()
    out.update(la_p=p.detach().numpy(), la_mask=amask.numpy(), la_out=la.detach().numpy(), la_dp=p.grad.numpy(),
               ld_d=d.numpy(), ld_mu=mu.detach().numpy(), ld_s2=s2.detach().numpy(), ld_mask=dmask.numpy(),
               ld_out=ld.detach().numpy(), ld_dmu=mu.grad.numpy(), ld_ds2=s2.grad.numpy(), w=w.numpy())
    np.savez_compressed(os.path.join(HERE, "rl.npz"), **out)
    print({k: (v.shape, v.dtype) for k, v in out.items()})
    print("pairs good same", out["pairs_good_same"], "accept", out["pairs_good_accept"], out["pairs_poor_accept"])
    print("gtpairs diff", out["gtpairs_diff"])


if __name__ == "__main__":
    main()
