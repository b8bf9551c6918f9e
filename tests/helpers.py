"""Shared helpers for the parity tests (oracle <-> golden <-> HIP)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        meta = json.load(f)
    data = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    return meta, data


def oracle_state(task, arch, seed, Hm=30, Wm=40, dtype=torch.float64, family="default"):
    from scanpaths_amd.procedural import procedural_state_dict
    from scanpaths_amd.spec import model_spec
    sd = procedural_state_dict(model_spec(task, arch, Hm, Wm), seed, family)
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}


def case_inputs(meta, dtype=torch.float64):
    from scanpaths_amd.synth import make_batch
    b = make_batch(meta["task"], meta["B"], meta["H"], meta["W"], meta["T"], seed=meta["seed"])
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in b.items()}


def max_err(a, b):
    if a is None:
        a = torch.zeros_like(torch.as_tensor(b))
    a = torch.as_tensor(a).detach().cpu().to(torch.float64)
    b = torch.as_tensor(b).detach().cpu().to(torch.float64)
    return (a - b).abs().max().item()


# ---- validation-metric goldens (tests/golden/make_golden_eval.py) -------------------------------------------------------------
FV_DTYPE = {"names": ("start_x", "start_y", "duration"), "formats": ("f8", "f8", "f8")}


def toy_multimatch(fv1, fv2, screensize):
    """A FIXED deterministic stand-in for multimatch_gaze.docomparison (third-party, absent here and not vendored by the
    reference): 5 values in (0, 1] from simple float64 statistics of the two scanpaths, NaN when either has fewer than 3 fixations
    (the package's rule, which is what makes the reference drop a pair, utils/evaluation.py:215-217).  The SAME function is the
    `multimatch_gaze` module of the golden run and the `multimatch=` argument of the build's functions, so the goldens pin
    everything around that column (pair enumeration, dropping, grouping, float32 collection, means / stds / best columns)."""
    import numpy as np
    if len(fv1) < 3 or len(fv2) < 3:
        return [float("nan")] * 5
    x1, y1, d1 = (np.asarray(fv1[k], dtype=np.float64) for k in ("start_x", "start_y", "duration"))
    x2, y2, d2 = (np.asarray(fv2[k], dtype=np.float64) for k in ("start_x", "start_y", "duration"))
    w, h = float(screensize[0]), float(screensize[1])
    return [float(1.0 / (1.0 + abs(x1.mean() - x2.mean()) / w)), float(1.0 / (1.0 + abs(y1.mean() - y2.mean()) / h)),
            float(min(len(x1), len(x2)) / max(len(x1), len(x2))), float(1.0 / (1.0 + abs(x1[0] - x2[0]) / w + abs(y1[0] - y2[0]) / h)),
            float(1.0 / (1.0 + abs(d1.sum() - d2.sum())))]


def unpack_scanpaths(fix, lens, counts=None):
    """[sum_len, 3] + per-scanpath lengths (+ per-image counts) -> (list of) lists of the reference's structured arrays"""
    import numpy as np
    out, o = [], 0
    for n in lens:
        a = np.zeros(int(n), dtype=FV_DTYPE)
        a["start_x"], a["start_y"], a["duration"] = fix[o:o + n, 0], fix[o:o + n, 1], fix[o:o + n, 2]
        out.append(a)
        o += int(n)
    if counts is None:
        return out
    grouped, k = [], 0
    for c in counts:
        grouped.append(out[k:k + int(c)])
        k += int(c)
    return grouped


EVAL_COLUMNS = [("MultiMatch", k) for k in ("vector", "direction", "length", "position", "duration")] + \
               [("ScanMatch", "w/o duration"), ("ScanMatch", "with duration"), ("VAME", "SED"), ("VAME", "STDE"),
                ("VAME", "SED_best"), ("VAME", "STDE_best")]
EVAL_CATEGORIES = ("all", "right_answer", "wrong_answer")


def metrics_table(d):
    """the reference's nested metric dict -> [3 categories, 11 columns] float64"""
    import numpy as np
    return np.array([[float(d[c][g][k]) for g, k in EVAL_COLUMNS] for c in EVAL_CATEGORIES], dtype=np.float64)


def oracle_bench_case(args):
    """(runs in a spawned worker process) the oracle's train step (loss + gradients) and eval forward of one AiR case in ONE dtype.
    args = (dtype name, seed, Hm, Wm, T, NB, H, W, threads) -> ({"train/<key>": array, "eval/<key>": array}, {param: grad}, loss)"""
    import torch
    from oracle import scanpath_oracle as O
    from scanpaths_amd.spec import is_buffer
    from scanpaths_amd.synth import make_batch
    dtname, seed, Hm, Wm, T, NB, H, W, threads = args
    torch.set_num_threads(threads)
    dt = getattr(torch, dtname)
    b = make_batch("AiR", NB, H, W, T, seed=seed)
    sd = oracle_state("AiR", "resnet50", seed, Hm, Wm, dtype=dt, family="tame")
    bd = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in b.items()}
    out = {}
    with torch.no_grad():
        ev = O.forward(sd, "AiR", bd["images"], bd["attention_maps"], training=False, T=T)
    for k, v in ev.items():
        out["eval/" + k] = v.double().numpy()
    for k, v in sd.items():
        if v.is_floating_point() and not is_buffer(k):
            v.requires_grad_(True)
    tr = O.forward(sd, "AiR", bd["images"], bd["attention_maps"], bd["performances"], training=True, T=T)
    loss, _, _ = O.supervised_loss(tr, bd)
    loss.backward()
    for k, v in tr.items():
        out["train/" + k] = v.detach().double().numpy()
    grads = {k: v.grad.numpy() for k, v in sd.items() if v.requires_grad and v.grad is not None}
    return out, grads, float(loss.detach())
