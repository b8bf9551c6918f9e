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


# ====================================================================================================================
# Validation / test metrics: evaluation_performance_related (utils/evaluation.py:188-359) and human_evaluation (:11-186).
# The reference runs, per (ground-truth, other) scanpath pair, MultiMatch (third-party multimatch_gaze==0.1.2, sp_baseline.yml:65,
# NOT vendored), two pure-python Needleman-Wunsch ScanMatch runs, SED and STDE inside nested loops.  Here the pairs of the whole
# call are collected first, ScanMatch (x2) / SED / STDE are scored by three batched device calls, and the per-image grouping,
# means / stds and the "best SED / STDE" columns are assembled exactly as the reference does.
# MultiMatch: ``multimatch`` = a callable docomparison(fv1, fv2, screensize=[320, 240]) -> 5 values.  Default: the installed
# multimatch_gaze if importable, else utils/evaltools/multimatch.py (a restatement of the published algorithm, parity unpinned).
# Quirk kept: the dict entry "w/o duration" holds column 5 = the score WITH duration and vice versa (:292-293 append the
# with-duration score first, :323-324 label them the other way round).
# ====================================================================================================================
def _default_multimatch():
    """the installed multimatch_gaze (the reference's dependency) if the user has it; else None = the batched device restatement
    (utils/evaltools/multimatch.multimatch_pairs: every pair of the call in one launch)"""
    try:
        import multimatch_gaze as mm
        return mm.docomparison
    except Exception:
        return None


def _multimatch_rows(mm, candidates):
    """candidates: list of (fixation vectors 1, fixation vectors 2) -> list of 5-value rows (NaNs where MultiMatch cannot score).
    mm None: one device launch for all candidates; else the per-pair callable (the reference's loop)."""
    if not candidates:
        return []
    if mm is not None:
        return [list(mm(a, b, screensize=[320, 240])) for a, b in candidates]
    from .evaltools.multimatch import multimatch_pairs
    paths, pairs = [], []
    for a, b in candidates:
        paths.extend([a, b])
        pairs.append((len(paths) - 2, len(paths) - 1))
    return [list(r) for r in multimatch_pairs(paths, pairs, [320, 240])]


def _rows_for_pairs(paths, pairs, sm_wd, sm_wod, mm_rows):
    """[npairs, 9] float64: 5 MultiMatch values, ScanMatch with duration, without duration, SED, STDE for (gt, other) pairs"""
    from .evaltools.visual_attention_metrics import sed_stde_pairs
    if not pairs:
        return np.zeros((0, 9))
    wd = _score_pairs(sm_wd, paths, pairs)
    wod = _score_pairs(sm_wod, paths, pairs)
    sed, stde = sed_stde_pairs(paths, pairs, (240, 320, 3))
    out = np.empty((len(pairs), 9), dtype=np.float64)
    out[:, :5] = np.asarray(mm_rows, dtype=np.float64).reshape(-1, 5)
    out[:, 5], out[:, 6] = wd, wod
    out[:, 7], out[:, 8] = sed.cpu().numpy(), stde.cpu().numpy()
    return out


def _summarise(collect_all, collect_right, collect_wrong, mean_name):
    keep = lambda lst: [a for a in lst if len(a) != 0]
    collected = [keep(collect_all), keep(collect_right), keep(collect_wrong)]
    summary_mean, summary_std = [], []
    for specific in collected:
        rl = np.concatenate(specific, axis=0)
        mean, std = rl.mean(0), rl.std(0)
        tmp = np.concatenate([np.concatenate([[a[:, 7].min(keepdims=True), a[:, 8].max(keepdims=True)]]).transpose((1, 0))
                              for a in specific], axis=0)
        summary_mean.append(np.concatenate([mean, tmp.mean(0)], axis=0))
        summary_std.append(np.concatenate([std, tmp.std(0)], axis=0))
    out = []
    for summ in (summary_mean, summary_std):
        d = dict()
        for category, v in zip(["all", "right_answer", "wrong_answer"], summ):
            d[category] = {"MultiMatch": {"vector": v[0], "direction": v[1], "length": v[2], "position": v[3], "duration": v[4]},
                           "ScanMatch": {"w/o duration": v[5], "with duration": v[6]},
                           "VAME": {"SED": v[7], "STDE": v[8], "SED_best": v[9], "STDE_best": v[10]}}
        out.append(d)
    return out[0], out[1]


def _make_scanmatch():
    from .evaltools.scanmatch import ScanMatch
    return (ScanMatch(Xres=320, Yres=240, Xbin=16, Ybin=12, Offset=(0, 0), TempBin=50, Threshold=3.5),
            ScanMatch(Xres=320, Yres=240, Xbin=16, Ybin=12, Offset=(0, 0), Threshold=3.5))


def evaluation_performance_related(gt_fix_vectors, predict_fix_vectors, all_performances, all_allocated_performances,
                                   multimatch=None):
    """(utils/evaluation.py:188-359)  -> cur_metrics, cur_metrics_std, scores_of_each_images"""
    mm = multimatch or _default_multimatch()
    sm_wd, sm_wod = _make_scanmatch()
    cand = [(gt_fix_vectors[index][inner], predict_fix_vectors[index], index, inner)
            for index in range(len(gt_fix_vectors)) for inner in range(len(gt_fix_vectors[index]))]
    mm_all = _multimatch_rows(mm, [(a, b) for a, b, _, _ in cand])
    paths, pairs, mm_rows, owner = [], [], [], []
    pred_slot = {}
    for (gt, pred, index, inner_index), rlt in zip(cand, mm_all):
        if index not in pred_slot:
            pred_slot[index] = len(paths)
            paths.append(_as_ms(pred))
        if np.any(np.isnan(np.asarray(rlt, dtype=np.float64))):
            continue                                                    # (:215-217) pairs MultiMatch cannot score are dropped
        paths.append(_as_ms(gt))
        pairs.append((len(paths) - 1, pred_slot[index]))
        mm_rows.append(rlt)
        owner.append((index, inner_index))
    rows = _rows_for_pairs(paths, pairs, sm_wd, sm_wod, mm_rows)
    collect_all, collect_right, collect_wrong, scores_of_each_images = [], [], [], []
    k = 0
    for index in range(len(gt_fix_vectors)):
        sample_all, sample_right, sample_wrong = [], [], []
        while k < len(owner) and owner[k][0] == index:
            inner_index = owner[k][1]
            r = list(rows[k])
            sample_all.append(r)
            if all_performances[index][inner_index] == True and all_allocated_performances[index] == True:      # noqa: E712
                sample_right.append(r)
            elif all_performances[index][inner_index] == False and all_allocated_performances[index] == False:  # noqa: E712
                sample_wrong.append(r)
            k += 1
        collect_all.append(np.array(sample_all, dtype=np.float32))
        collect_right.append(np.array(sample_right, dtype=np.float32))
        collect_wrong.append(np.array(sample_wrong, dtype=np.float32))
        chosen = sample_right if all_allocated_performances[index] == True else sample_wrong                   # noqa: E712
        scores_of_each_images.append(list(np.array(chosen).mean(axis=0)) if chosen != [] else list(np.zeros((9,), dtype=np.float64)))
    cur_metrics, cur_metrics_std = _summarise(collect_all, collect_right, collect_wrong, "cur")
    return cur_metrics, cur_metrics_std, scores_of_each_images


def human_evaluation(dataloader, multimatch=None):
    """(utils/evaluation.py:11-186)  every ordered pair of distinct human scanpaths of an image; dataloader yields batches with
    "fix_vectors", "performances", "question_ids"  -> human_metrics, human_metrics_std, scores_of_each_images_dict"""
    mm = multimatch or _default_multimatch()
    sm_wd, sm_wod = _make_scanmatch()
    paths, images, gt_qid_name, cand = [], [], [], []
    for batch in dataloader:
        gt_qid_name.extend(batch["question_ids"])
        for fix_vectors, performances in zip(batch["fix_vectors"], batch["performances"]):
            img = len(images)
            images.append(performances)
            base = len(paths)
            for fv in fix_vectors:
                paths.append(_as_ms(fv))
            for index_1 in range(len(fix_vectors)):
                for index_2 in range(len(fix_vectors)):
                    if index_2 != index_1:
                        cand.append((fix_vectors[index_1], fix_vectors[index_2], base + index_1, base + index_2, (img, index_1, index_2)))
    mm_all = _multimatch_rows(mm, [(a, b) for a, b, _, _, _ in cand])
    pairs, mm_rows, owner = [], [], []
    for (_, _, p1, p2, own), rlt in zip(cand, mm_all):
        if np.any(np.isnan(np.asarray(rlt, dtype=np.float64))):
            continue
        pairs.append((p1, p2))
        mm_rows.append(rlt)
        owner.append(own)
    rows = _rows_for_pairs(paths, pairs, sm_wd, sm_wod, mm_rows)
    collect_all, collect_right, collect_wrong, good_scores, poor_scores = [], [], [], [], []
    k = 0
    for img, performances in enumerate(images):
        sample_all, sample_right, sample_wrong = [], [], []
        while k < len(owner) and owner[k][0] == img:
            _, i1, i2 = owner[k]
            r = list(rows[k])
            sample_all.append(r)
            if performances[i1] == True and performances[i2] == True:        # noqa: E712
                sample_right.append(r)
            elif performances[i1] == False and performances[i2] == False:    # noqa: E712
                sample_wrong.append(r)
            k += 1
        collect_all.append(np.array(sample_all, dtype=np.float32))
        collect_right.append(np.array(sample_right, dtype=np.float32))
        collect_wrong.append(np.array(sample_wrong, dtype=np.float32))
        good_scores.append(list(np.array(sample_right, dtype=np.float64).mean(axis=0)) if sample_right != [] else list(np.zeros((9,))))
        poor_scores.append(list(np.array(sample_wrong, dtype=np.float64).mean(axis=0)) if sample_wrong != [] else list(np.zeros((9,))))
    human_metrics, human_metrics_std = _summarise(collect_all, collect_right, collect_wrong, "human")
    scores = {name: dict([(True, g), (False, p)]) for name, g, p in zip(gt_qid_name, good_scores, poor_scores)}
    return human_metrics, human_metrics_std, scores
