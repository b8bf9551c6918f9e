"""Golden vectors for the OSIE / COCO_Search18 forms of validation scoring and of the RL (self-critical) step -- SURVEY.md §8 rows
f2 / f3 widened (VERDICT r4 "next" #7) -- from the REAL reference, build container only:

    python tests/golden/make_golden_eval_tasks.py            (runs itself once per task in a child process: both trees call their
                                                              package `utils`, one interpreter can import only one of them)

Per task it imports /root/reference/<task>/utils/evaluation.py where it lies and runs
  * evaluation(gt_fix_vectors, predict_fix_vectors)                         OSIE :151-282, COCO_Search18 :180-311
  * human_evaluation(dataloader)                                            OSIE :11-148 (equal scanpath counts), COCO_Search18 :11-178 (ragged)
  * pairs_eval (OSIE :284-340, 11 columns)  /  pairs_eval_scanmatch (COCO_Search18 :313-352, 2 columns)
on seeded scanpath sets, and executes the reward / baseline / loss lines of the task's train.py (OSIE/train.py:248-258,
COCO_Search18/train.py:269-279: read from the reference file at generation time and exec'd on seeded tensors -- nothing of it is
stored) with gradients.  Inputs + outputs go to tests/golden/eval_osie.npz / eval_coco.npz.

Shims: as tests/golden/make_golden_eval.py (multimatch_gaze <- tests/helpers.py::toy_multimatch, tqdm no-op, cv2 / matplotlib empty)."""
import os
import subprocess
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
TASKS = {"osie": ("OSIE", "eval_osie.npz"), "coco": ("COCO_Search18", "eval_coco.npz")}
# the reward -> loss lines of the RL branch: (first line, last line, text the first line must contain) in <task>/train.py
RL_LINES = {"osie": (248, 258, "neg_log_actions_tensor = torch.cat(neg_log_actions_batch, dim=0)"),
            "coco": (269, 279, "neg_log_actions_tensor = torch.cat(neg_log_actions_batch, dim=0)")}


def scanpath(g, n=None):
    from helpers import FV_DTYPE
    n = int(g.integers(1, 12)) if n is None else n
    fv = np.zeros(n, dtype=FV_DTYPE)
    fv["start_x"], fv["start_y"] = g.uniform(0, 320, n), g.uniform(0, 240, n)
    fv["duration"] = g.uniform(0.05, 0.9, n)
    return fv


def flatten(paths):
    return (np.concatenate([np.stack([f["start_x"], f["start_y"], f["duration"]], 1) for f in paths], 0),
            np.array([len(f) for f in paths]))


def table(d):
    """the nested metric dict of evaluation / human_evaluation (no categories) -> [11] float64 in helpers.EVAL_COLUMNS order"""
    from helpers import EVAL_COLUMNS
    return np.array([float(d[g][k]) for g, k in EVAL_COLUMNS], dtype=np.float64)


def run(task_key):
    import torch
    from helpers import toy_multimatch
    task, fname = TASKS[task_key]
    for name in ("multimatch_gaze", "tqdm", "cv2", "matplotlib", "matplotlib.pyplot"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["multimatch_gaze"].docomparison = toy_multimatch

    class _NoBar:
        def __init__(self, *a, **k):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def update(self, n=1):
            pass

    sys.modules["tqdm"].tqdm = _NoBar
    sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
    sys.path.insert(0, f"/root/reference/{task}")
    import utils.evaluation as REF
    from utils.evaltools.scanmatch import ScanMatch
    g = np.random.Generator(np.random.PCG64(501 if task_key == "osie" else 502))
    out = {}
    # ---- evaluation(): 6 images x 4 human scanpaths (the function reshapes SED / STDE to [-1, len(last image's list)]); short
    #      scanpaths (< 3 fixations: MultiMatch NaN rows, eliminated from the MultiMatch means only), a 2-fixation prediction ------
    counts = [4] * 6
    gt = [[scanpath(g) for _ in range(c)] for c in counts]
    gt[1][2] = scanpath(g, 2)
    gt[4][0] = scanpath(g, 1)
    pred = [scanpath(g, int(g.integers(3, 14))) for _ in counts]
    pred[3] = scanpath(g, 2)
    cur, cur_std, scores = REF.evaluation(gt, pred)
    out["ev_gt_fix"], out["ev_gt_len"] = flatten([f for l in gt for f in l])
    out["ev_gt_count"] = np.array(counts)
    out["ev_pred_fix"], out["ev_pred_len"] = flatten(pred)
    out["ev_mean"], out["ev_std"] = table(cur), table(cur_std)
    out["ev_scores"] = np.array(scores, dtype=np.float64)
    # ---- human_evaluation(): OSIE reshapes to [-1, len - 1] (equal counts), COCO_Search18 keeps per-image ranges (ragged counts) ---
    hcounts = [4, 4, 4, 4, 4] if task_key == "osie" else [4, 3, 2, 5, 3]
    hfix = [[scanpath(g, int(g.integers(3, 11))) for _ in range(c)] for c in hcounts]
    hfix[2][1] = scanpath(g, 2)          # a pair MultiMatch cannot score (its other columns still count)
    names = [f"img{i:03d}.jpg" for i in range(len(hcounts))]
    loader = [{"fix_vectors": hfix[:2], "img_names": names[:2]}, {"fix_vectors": hfix[2:], "img_names": names[2:]}]
    hm, hstd, hscores = REF.human_evaluation(loader)
    out["hum_fix"], out["hum_len"] = flatten([f for l in hfix for f in l])
    out["hum_count"] = np.array(hcounts)
    out["hum_mean"], out["hum_std"] = table(hm), table(hstd)
    out["hum_scores"] = np.array([hscores[n] for n in names], dtype=np.float64)
    # (OSIE's human_evaluation does not eliminate NaN rows: one unscorable pair makes its five MultiMatch means NaN -- kept, as the
    # reference; a second set without short scanpaths pins those columns)
    h2counts = [3, 3, 3] if task_key == "osie" else [3, 2, 4]
    h2fix = [[scanpath(g, int(g.integers(3, 11))) for _ in range(c)] for c in h2counts]
    names2 = [f"set2_{i:03d}.jpg" for i in range(len(h2counts))]
    hm2, hstd2, hscores2 = REF.human_evaluation([{"fix_vectors": h2fix, "img_names": names2}])
    out["hum2_fix"], out["hum2_len"] = flatten([f for l in h2fix for f in l])
    out["hum2_count"] = np.array(h2counts)
    out["hum2_mean"], out["hum2_std"] = table(hm2), table(hstd2)
    out["hum2_scores"] = np.array([hscores2[n] for n in names2], dtype=np.float64)
    # ---- the reward of the RL branch: per image the mean over its human scanpaths (divided by their FULL count even after NaN rows
    #      were eliminated -- kept), NaN row for an image with nothing left --------------------------------------------------------
    wd = ScanMatch(Xres=320, Yres=240, Xbin=16, Ybin=12, Offset=(0, 0), TempBin=50, Threshold=3.5)
    wod = ScanMatch(Xres=320, Yres=240, Xbin=16, Ybin=12, Offset=(0, 0), Threshold=3.5)
    pcounts = [4, 3, 1, 5, 2, 2]
    pgt = [[scanpath(g, int(g.integers(3, 12))) for _ in range(c)] for c in pcounts]
    pgt[0][1] = scanpath(g, 2)           # OSIE: NaN MultiMatch row -> eliminated, mean still divided by 4
    pgt[2][0] = scanpath(g, 1)           # OSIE: the image's only row is NaN -> NaN reward row (the sample is redrawn in train.py)
    ppred = [scanpath(g, int(g.integers(3, 12))) for _ in pcounts]
    fn = REF.pairs_eval if task_key == "osie" else REF.pairs_eval_scanmatch
    rew = fn(pgt, ppred, wd, wod)
    out["pe_gt_fix"], out["pe_gt_len"] = flatten([f for l in pgt for f in l])
    out["pe_gt_count"] = np.array(pcounts)
    out["pe_pred_fix"], out["pe_pred_len"] = flatten(ppred)
    out["pe_reward"] = np.asarray(rew, dtype=np.float64)
    # ---- reward -> baseline -> loss, the reference's own lines, with gradients -------------------------------------------------
    lo, hi, must = RL_LINES[task_key]
    src = open(f"/root/reference/{task}/train.py").read().split("\n")[lo - 1:hi]
    assert must in src[0] and "loss = loss_actions + loss_duration" in src[-1], (src[0], src[-1])
    indent = len(src[0]) - len(src[0].lstrip())
    code = "\n".join(l[indent:] for l in src)
    tg = torch.Generator().manual_seed(17)
    S, N, C = 5, 6, (11 if task_key == "osie" else 2)
    nla = [torch.rand(1, N, generator=tg, dtype=torch.float64).requires_grad_(True) for _ in range(S)]
    nld = [torch.rand(1, N, generator=tg, dtype=torch.float64).requires_grad_(True) for _ in range(S)]
    mr = [(torch.rand(1, N, C, generator=tg) * 0.9 + 0.05).to(torch.float32) for _ in range(S)]
    torch.Tensor.get_device = lambda self: self.device          # (CPU tensors: .to(get_device()) of the reference's lines)
    import scipy.stats
    ns = {"torch": torch, "scipy": scipy, "np": np, "neg_log_actions_batch": nla, "neg_log_durations_batch": nld,
          "metrics_reward_batch": mr}
    exec(code, ns)
    ns["loss"].backward()
    out["rl_nla"] = torch.cat(nla, 0).detach().numpy()
    out["rl_nld"] = torch.cat(nld, 0).detach().numpy()
    out["rl_reward"] = torch.cat(mr, 0).numpy()
    out["rl_hmean"] = np.asarray(ns["metrics_reward_hmean"], dtype=np.float64)
    out["rl_loss"] = np.array(float(ns["loss"].detach()))
    out["rl_loss_actions"] = np.array(float(ns["loss_actions"].detach()))
    out["rl_loss_duration"] = np.array(float(ns["loss_duration"].detach()))
    out["rl_dnla"] = torch.cat([t.grad for t in nla], 0).numpy()
    out["rl_dnld"] = torch.cat([t.grad for t in nld], 0).numpy()
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(task, {k: (v.shape, str(v.dtype)) for k, v in out.items()})
    print(task, "evaluation mean", out["ev_mean"])
    print(task, "pairs reward", out["pe_reward"])
    print(task, "rl loss", out["rl_loss"])


if __name__ == "__main__":
    import warnings
    if len(sys.argv) > 1:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            run(sys.argv[1])
    else:
        for key in TASKS:
            subprocess.run([sys.executable, os.path.abspath(__file__), key], check=True)
