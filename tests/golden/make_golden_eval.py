"""Golden vectors for the validation metrics from the REAL reference, build container only:

    python tests/golden/make_golden_eval.py

Runs /root/reference/AiR/utils/evaluation.py::evaluation_performance_related (:188-359) and ::human_evaluation (:11-186) on
seeded scanpath sets and stores inputs + outputs in tests/golden/eval_metrics.npz.

Shims (nothing of the reference is copied; it is imported where it lies):
  * `multimatch_gaze` (third-party, multimatch_gaze==0.1.2, sp_baseline.yml:65, absent and not vendored) <- a module whose
    `docomparison` is tests/helpers.py::toy_multimatch, a fixed deterministic function (NaN below 3 fixations);
  * `tqdm` <- a no-op context manager; cv2 / matplotlib <- empty modules (imported, unused by these functions);
  * numpy: the reference is pinned to numpy==1.19.2, where `ndarray != []` with mismatched shapes evaluates to the scalar True
    (DeprecationWarning) and an EMPTY array compared with [] is falsy -- the idiom `[_ for _ in rlts if _ != []]` (:108-110,
    :279-281) drops the empty per-image arrays.  numpy 2.2 raises there.  The module's `np` is therefore replaced by a proxy whose
    `np.array` returns an ndarray subclass that answers `!= []` the numpy-1.19 way; everything else is plain numpy."""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from helpers import FV_DTYPE, metrics_table, toy_multimatch      # noqa: E402

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
sys.path.insert(0, "/root/reference/AiR")
import utils.evaluation as REF      # noqa: E402


class _Arr119(np.ndarray):
    def __ne__(self, other):
        if isinstance(other, list) and len(other) == 0:
            return self.size != 0          # numpy 1.19.2: shape mismatch -> True; empty vs [] -> empty (falsy) array
        return np.ndarray.__ne__(self, other)


class _NumpyProxy:
    def __getattr__(self, k):
        return getattr(np, k)

    @staticmethod
    def array(*a, **k):
        return np.array(*a, **k).view(_Arr119)


REF.np = _NumpyProxy()


def scanpath(g, n=None):
    n = int(g.integers(1, 12)) if n is None else n
    fv = np.zeros(n, dtype=FV_DTYPE)
    fv["start_x"], fv["start_y"] = g.uniform(0, 320, n), g.uniform(0, 240, n)
    fv["duration"] = g.uniform(0.05, 0.9, n)
    return fv


def flatten(paths):
    return (np.concatenate([np.stack([f["start_x"], f["start_y"], f["duration"]], 1) for f in paths], 0),
            np.array([len(f) for f in paths]))


def main():
    g = np.random.Generator(np.random.PCG64(2024))
    out = {}
    # ---- evaluation_performance_related: 9 images; short scanpaths (< 3 fixations: dropped pairs), an image whose prediction is
    #      too short (all of its pairs dropped -> empty arrays filtered by the `!= []` idiom), all-good / all-poor images --------
    n_img = 9
    counts = [4, 3, 2, 5, 1, 3, 4, 2, 3]
    gt = [[scanpath(g) for _ in range(c)] for c in counts]
    gt[3][1] = scanpath(g, 2)
    gt[6][0] = scanpath(g, 1)
    perf = [[bool(g.random() < 0.5) for _ in range(c)] for c in counts]
    perf[0], perf[1] = [True] * counts[0], [False] * counts[1]
    pred = [scanpath(g, int(g.integers(3, 14))) for _ in range(n_img)]
    pred[2] = scanpath(g, 2)
    alloc = [True, False, True, True, False, True, False, False, True]
    cur, cur_std, scores = REF.evaluation_performance_related(gt, pred, perf, alloc)
    out["epr_gt_fix"], out["epr_gt_len"] = flatten([f for l in gt for f in l])
    out["epr_gt_count"] = np.array(counts)
    out["epr_perf"] = np.array([int(p) for l in perf for p in l])
    out["epr_pred_fix"], out["epr_pred_len"] = flatten(pred)
    out["epr_alloc"] = np.array([int(a) for a in alloc])
    out["epr_mean"], out["epr_std"] = metrics_table(cur), metrics_table(cur_std)
    out["epr_scores"] = np.array(scores, dtype=np.float64)
    # ---- human_evaluation: 2 batches of 3 images ---------------------------------------------------------------------------------
    hcounts = [4, 3, 2, 5, 3, 4]
    hfix = [[scanpath(g, int(g.integers(2, 11))) for _ in range(c)] for c in hcounts]
    hperf = [[bool(g.random() < 0.55) for _ in range(c)] for c in hcounts]
    hperf[1] = [True] * hcounts[1]
    qids = [f"q{i:03d}" for i in range(len(hcounts))]
    loader = [{"fix_vectors": hfix[:3], "performances": hperf[:3], "question_ids": qids[:3]},
              {"fix_vectors": hfix[3:], "performances": hperf[3:], "question_ids": qids[3:]}]
    hm, hstd, hscores = REF.human_evaluation(loader)
    out["hum_fix"], out["hum_len"] = flatten([f for l in hfix for f in l])
    out["hum_count"] = np.array(hcounts)
    out["hum_perf"] = np.array([int(p) for l in hperf for p in l])
    out["hum_mean"], out["hum_std"] = metrics_table(hm), metrics_table(hstd)
    out["hum_good"] = np.array([hscores[q][True] for q in qids], dtype=np.float64)
    out["hum_poor"] = np.array([hscores[q][False] for q in qids], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "eval_metrics.npz"), **out)
    print({k: (v.shape, str(v.dtype)) for k, v in out.items()})
    print("epr mean[all]", out["epr_mean"][0])
    print("hum mean[right]", out["hum_mean"][1])


if __name__ == "__main__":
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        main()
