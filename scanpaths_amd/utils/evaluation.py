"""ScanMatch reward glue of the RL (self-critical) phase with the reference's interface (utils/evaluation.py:361-576),
SURVEY.md §8 row f3.  The reference scores every (ground-truth scanpath, other scanpath) pair with two pure-python
Needleman-Wunsch runs inside nested loops; here the pairs of a whole batch are collected first and scored by ONE batched
device call per ScanMatch object (csrc/scanmatch.hip: one wavefront per pair), then grouped exactly as the reference does
(mean over the non-NaN rows of a group, NaN for an empty group, accept_flag False when NaN elimination empties a group).

    same, diff, accept = pairs_eval_scanmatch_performance_related(gt_fix_vectors, predict_fix_vectors, sm_wd, sm_wod,
                                                                   performance, given_performance)
    good, poor, good_vs_poor = gtpairs_eval_scanmatch_performance_related(gt_fix_vectors, sm_wd, sm_wod, performance)

Fixation vectors are the reference's structured arrays (start_x, start_y, duration in seconds) or plain [n, 3] arrays;
durations are converted to milliseconds as in the reference.  Result columns: [0] without duration, [1] with duration.
Any object with the reference's ScanMatch methods works; objects that also offer ``sequences`` / ``match_pairs`` (the HIP
ScanMatch) are driven in batched form."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def _as_ms(fv) -> np.ndarray:
    a = np.array([list(_) for _ in list(fv)], dtype=np.float64).reshape(-1, 3)
    a[:, -1] *= 1000
    return a


def _score_pairs(sm, paths: List[np.ndarray], pairs: Sequence[Tuple[int, int]]) -> np.ndarray:
    if not pairs:
        return np.zeros(0)
    if hasattr(sm, "match_pairs") and hasattr(sm, "sequences"):
        import torch
        seq, ln = sm.sequences(paths)
        return sm.match_pairs(seq, ln, seq, ln, torch.tensor(list(pairs), dtype=torch.int32)).cpu().numpy()
    seqs = [sm.fixationToSequence(p).astype(np.int32) for p in paths]
    return np.array([sm.match(seqs[i], seqs[j])[0] for i, j in pairs], dtype=np.float64)


def _group_mean(rows: np.ndarray, is_eliminating_nan: bool):
    """rows [k, 2] of one group -> (metric [2], emptied_by_nan)"""
    emptied = False
    if is_eliminating_nan and rows.shape[0] != 0:
        rows = rows[np.isnan(rows.sum(axis=1)) == False]          # noqa: E712  (as the reference)
        emptied = rows.shape[0] == 0
    if rows.shape[0] != 0:
        return np.sum(rows, axis=0) / rows.shape[0], emptied
    return np.array([np.nan] * 2), emptied


def pairs_eval_scanmatch_performance_related(gt_fix_vectors, predict_fix_vectors, ScanMatchwithDuration,
                                             ScanMatchwithoutDuration, performance, given_performance,
                                             is_eliminating_nan=True):
    """(utils/evaluation.py:361-422) per image: mean ScanMatch of the prediction against the ground-truth scanpaths whose
    performance equals ``given_performance`` (same) and against the others (diff)."""
    paths, pairs, owner = [], [], []
    for index in range(len(gt_fix_vectors)):
        pi = len(paths)
        paths.append(_as_ms(predict_fix_vectors[index]))
        for inner_index in range(len(gt_fix_vectors[index])):
            paths.append(_as_ms(gt_fix_vectors[index][inner_index]))
            pairs.append((len(paths) - 1, pi))                     # match(gt, prediction), as the reference
            owner.append((index, performance[index][inner_index] == given_performance))
    wd = _score_pairs(ScanMatchwithDuration, paths, pairs)
    wod = _score_pairs(ScanMatchwithoutDuration, paths, pairs)
    accept_flag = True
    same_out, diff_out = [], []
    for index in range(len(gt_fix_vectors)):
        same = np.array([[wod[k], wd[k]] for k, (i, s) in enumerate(owner) if i == index and s]).reshape(-1, 2)
        diff = np.array([[wod[k], wd[k]] for k, (i, s) in enumerate(owner) if i == index and not s]).reshape(-1, 2)
        m_same, e1 = _group_mean(same, is_eliminating_nan)
        m_diff, e2 = _group_mean(diff, is_eliminating_nan)
        if e1 or e2:
            accept_flag = False
        same_out.append(m_same)
        diff_out.append(m_diff)
    return np.array(same_out), np.array(diff_out), accept_flag


def gtpairs_eval_scanmatch_performance_related(gt_fix_vectors, ScanMatchwithDuration, ScanMatchwithoutDuration, performance,
                                               is_eliminating_nan=True):
    """(utils/evaluation.py:425-576) per image: mean ScanMatch among the good-performance ground-truth scanpaths, among the
    poor ones, and between the two groups (the last only when BOTH groups have more than one member, as the reference)."""
    paths, pairs, owner = [], [], []
    for index, (gt_fix_vector, performance_val) in enumerate(zip(gt_fix_vectors, performance)):
        base = len(paths)
        for fv in gt_fix_vector:
            paths.append(_as_ms(fv))
        good = [base + k for k in range(len(performance_val)) if performance_val[k] == True]       # noqa: E712
        poor = [base + k for k in range(len(performance_val)) if not performance_val[k] == True]   # noqa: E712
        for grp, tag in ((good, "good"), (poor, "poor")):
            if len(grp) > 1:
                for a in range(len(grp)):
                    for b in range(a + 1, len(grp)):
                        pairs.append((grp[a], grp[b]))
                        owner.append((index, tag))
        if len(good) > 1 and len(poor) > 1:
            for a in good:
                for b in poor:
                    pairs.append((a, b))
                    owner.append((index, "diff"))
    wd = _score_pairs(ScanMatchwithDuration, paths, pairs)
    wod = _score_pairs(ScanMatchwithoutDuration, paths, pairs)
    out = {"good": [], "poor": [], "diff": []}
    for index in range(len(gt_fix_vectors)):
        for tag in out:
            rows = np.array([[wod[k], wd[k]] for k, (i, t) in enumerate(owner) if i == index and t == tag]).reshape(-1, 2)
            out[tag].append(_group_mean(rows, is_eliminating_nan)[0])
    return np.array(out["good"]), np.array(out["poor"]), np.array(out["diff"])
