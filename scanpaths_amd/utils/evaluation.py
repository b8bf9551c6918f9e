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


# ====================================================================================================================
# OSIE / COCO_Search18 forms (SURVEY.md §8 rows f2 / f3 widened).  Free-viewing / visual-search scanpaths carry no answer
# correctness, so there is no performance split: ONE metric set per call instead of all / right / wrong.
#   evaluation(gt, pred)             OSIE/utils/evaluation.py:151-282 == COCO_Search18/utils/evaluation.py:180-311 (identical code)
#   human_evaluation(loader, task=)  OSIE :11-148 (SED / STDE reshaped to [-1, n - 1]: equal scanpath counts; MultiMatch NaN rows NOT
#                                    eliminated) / COCO_Search18 :11-178 (per-image ranges: ragged counts; NaN rows eliminated)
#   pairs_eval                       OSIE :284-340  (RL reward, 11 columns)
#   pairs_eval_scanmatch             COCO_Search18 :313-352 (RL reward, 2 columns)
# As above: the pairs of a whole call are scored by batched device launches (ScanMatch x2, SED / STDE, MultiMatch), the grouping and the
# statistics follow the reference line by line -- including (a) the per-pair rows hold the score WITH duration in column 5 and the one
# WITHOUT in column 6 while the dicts are labelled correctly here (unlike AiR's :323-324), (b) pairs_eval divides an image's column sums
# by the FULL number of its human scanpaths even after NaN rows were eliminated (:329), (c) pairs_eval scores ScanMatch / SED / STDE only
# for the pairs MultiMatch could score.
# ====================================================================================================================
def _flat_metrics(mm_mean, mm_std, wd, wod, sed_all, stde_all, sed_best, stde_best):
    mean = {"MultiMatch": dict(zip(("vector", "direction", "length", "position", "duration"), mm_mean)),
            "ScanMatch": {"w/o duration": np.mean(wod), "with duration": np.mean(wd)},
            "VAME": {"SED": sed_all.mean(), "STDE": stde_all.mean(), "SED_best": sed_best.mean(), "STDE_best": stde_best.mean()}}
    std = {"MultiMatch": dict(zip(("vector", "direction", "length", "position", "duration"), mm_std)),
           "ScanMatch": {"w/o duration": np.std(wod), "with duration": np.std(wd)},
           "VAME": {"SED": sed_all.std(), "STDE": stde_all.std(), "SED_best": sed_best.std(), "STDE_best": stde_best.std()}}
    return mean, std


def _score_all(paths, pairs, mm_rows, sm_wd=None, sm_wod=None):
    """[npairs, 9] float64 rows (5 MultiMatch, ScanMatch with / without duration, SED, STDE) for (first, second) path-index pairs"""
    if sm_wd is None:
        sm_wd, sm_wod = _make_scanmatch()
    return _rows_for_pairs(paths, pairs, sm_wd, sm_wod, mm_rows)


def evaluation(gt_fix_vectors, predict_fix_vectors, is_eliminating_nan=True, multimatch=None):
    """(OSIE/utils/evaluation.py:151-282, COCO_Search18/utils/evaluation.py:180-311) -> cur_metrics, cur_metrics_std, scores_of_each_images.
    Every (human scanpath of the image, prediction) pair; like the reference, SED / STDE are regrouped as [-1, number of human
    scanpaths of the LAST image] for the "best" columns (equal counts per image expected)."""
    mm = multimatch or _default_multimatch()
    paths, pairs, cand, per_image = [], [], [], []
    for index in range(len(gt_fix_vectors)):
        pi = len(paths)
        paths.append(_as_ms(predict_fix_vectors[index]))
        per_image.append(len(gt_fix_vectors[index]))
        for inner in gt_fix_vectors[index]:
            paths.append(_as_ms(inner))
            pairs.append((len(paths) - 1, pi))
            cand.append((inner, predict_fix_vectors[index]))
    mm_rows = _multimatch_rows(mm, cand)
    rows = _score_all(paths, pairs, mm_rows)
    scores_of_each_images, k = [], 0
    for n in per_image:
        scores_of_each_images.append(list(np.array([list(r) for r in rows[k:k + n]]).mean(axis=0)))
        k += n
    mmr = rows[:, :5]
    if is_eliminating_nan:
        mmr = mmr[np.isnan(mmr.sum(axis=1)) == False]          # noqa: E712
    sed = rows[:, 7].reshape(-1, per_image[-1])
    stde = rows[:, 8].reshape(-1, per_image[-1])
    mean, std = _flat_metrics(np.mean(mmr, axis=0), np.std(mmr, axis=0), rows[:, 5], rows[:, 6], sed, stde, sed.min(-1), stde.max(-1))
    return mean, std, scores_of_each_images


def human_evaluation_free_viewing(dataloader, task="OSIE", multimatch=None):
    """human_evaluation of OSIE (:11-148) / COCO_Search18 (:11-178): every ordered pair of distinct human scanpaths of an image;
    batches carry "fix_vectors" and "img_names" -> human_metrics, human_metrics_std, {image name: mean score row of the image}"""
    assert task in ("OSIE", "COCO_Search18"), task
    mm = multimatch or _default_multimatch()
    paths, pairs, cand, names, groups = [], [], [], [], []       # groups: per image, per first scanpath: number of pairs (n - 1)
    for batch in dataloader:
        names.extend(batch["img_names"])
        for fix_vectors in batch["fix_vectors"]:
            base = len(paths)
            for fv in fix_vectors:
                paths.append(_as_ms(fv))
            n = len(fix_vectors)
            groups.append(n)
            for i1 in range(n):
                for i2 in range(n):
                    if i2 != i1:
                        pairs.append((base + i1, base + i2))
                        cand.append((fix_vectors[i1], fix_vectors[i2]))
    rows = _score_all(paths, pairs, _multimatch_rows(mm, cand))
    scores, k = [], 0
    for n in groups:
        cnt = n * (n - 1)
        scores.append(list(np.array([list(r) for r in rows[k:k + cnt]]).mean(axis=0)))
        k += cnt
    mmr = rows[:, :5]
    if task == "COCO_Search18":
        mmr = mmr[np.isnan(mmr.sum(axis=1)) == False]          # noqa: E712  (:79; OSIE keeps the NaN rows, its means turn NaN)
        sed_best, stde_best, k = [], [], 0
        for n in groups:                                        # (:88-125) best over ALL ordered pairs of the image
            cnt = n * (n - 1)
            sed_best.append(rows[k:k + cnt, 7].min())
            stde_best.append(rows[k:k + cnt, 8].max())
            k += cnt
        sed_all, stde_all = rows[:, 7], rows[:, 8]
        sed_best, stde_best = np.array(sed_best), np.array(stde_best)
    else:
        sed_all = rows[:, 7].reshape(-1, groups[-1] - 1)       # (:86-87) one row per first scanpath: best over its n - 1 partners
        stde_all = rows[:, 8].reshape(-1, groups[-1] - 1)
        sed_best, stde_best = sed_all.min(-1), stde_all.max(-1)
    mean, std = _flat_metrics(np.mean(mmr, axis=0), np.std(mmr, axis=0), rows[:, 5], rows[:, 6], sed_all, stde_all, sed_best, stde_best)
    return mean, std, dict(zip(names, scores))


def pairs_eval(gt_fix_vectors, predict_fix_vectors, ScanMatchwithDuration, ScanMatchwithoutDuration, is_eliminating_nan=True,
               multimatch=None):
    """(OSIE/utils/evaluation.py:284-340) RL reward of one sampled scanpath per image -> [N, 11] float: the 5 MultiMatch means, ScanMatch
    without / with duration, SED, STDE (column sums over the scorable pairs divided by the image's FULL number of human scanpaths),
    best SED (min) and best STDE (max); a NaN row where nothing was scorable (OSIE/train.py:236-238 then redraws the sample)."""
    mm = multimatch or _default_multimatch()
    cand = [(gt, predict_fix_vectors[index]) for index in range(len(gt_fix_vectors)) for gt in gt_fix_vectors[index]]
    mm_all = _multimatch_rows(mm, cand)
    paths, pairs, mm_rows, owner, k = [], [], [], [], 0
    for index in range(len(gt_fix_vectors)):
        pi = len(paths)
        paths.append(_as_ms(predict_fix_vectors[index]))
        for gt in gt_fix_vectors[index]:
            rlt = mm_all[k]
            k += 1
            if np.any(np.isnan(np.asarray(rlt, dtype=np.float64))):
                owner.append((index, None))                     # row of NaNs (:296-299): ScanMatch / SED / STDE are not run for it
                continue
            paths.append(_as_ms(gt))
            pairs.append((len(paths) - 1, pi))
            mm_rows.append(rlt)
            owner.append((index, len(pairs) - 1))
    rows = _rows_for_pairs(paths, pairs, ScanMatchwithDuration, ScanMatchwithoutDuration, mm_rows)
    out = []
    for index in range(len(gt_fix_vectors)):
        coll = []
        for i, slot in owner:
            if i != index:
                continue
            if slot is None:
                coll.append([np.nan] * 9)
            else:
                r = rows[slot]
                coll.append(list(r[:5]) + [r[6], r[5], r[7], r[8]])          # (:323) [w/o duration, with duration, SED, STDE]
        coll = np.array(coll, dtype=np.float64).reshape(-1, 9)
        if is_eliminating_nan:
            coll = coll[np.isnan(coll.sum(axis=1)) == False]   # noqa: E712
        if coll.shape[0] != 0:
            metric_mean = np.sum(coll, axis=0) / len(gt_fix_vectors[index])
            v = np.zeros((11,), dtype=np.float32)
            v[:9] = metric_mean[:9]
            v[9], v[10] = coll[:, 7].min(), coll[:, 8].max()
        else:
            v = np.array([np.nan] * 11)
        out.append(v)
    return np.array(out)


def pairs_eval_scanmatch(gt_fix_vectors, predict_fix_vectors, ScanMatchwithDuration, ScanMatchwithoutDuration, is_eliminating_nan=True):
    """(COCO_Search18/utils/evaluation.py:313-352) RL reward -> [N, 2]: per image the ScanMatch score without / with duration of the
    sampled scanpath against its human scanpaths, column sums over the non-NaN rows divided by the FULL number of human scanpaths."""
    paths, pairs, owner = [], [], []
    for index in range(len(gt_fix_vectors)):
        pi = len(paths)
        paths.append(_as_ms(predict_fix_vectors[index]))
        for gt in gt_fix_vectors[index]:
            paths.append(_as_ms(gt))
            pairs.append((len(paths) - 1, pi))
            owner.append(index)
    wd = _score_pairs(ScanMatchwithDuration, paths, pairs)
    wod = _score_pairs(ScanMatchwithoutDuration, paths, pairs)
    out = []
    for index in range(len(gt_fix_vectors)):
        coll = np.array([[wod[k], wd[k]] for k, i in enumerate(owner) if i == index], dtype=np.float64).reshape(-1, 2)
        if is_eliminating_nan:
            coll = coll[np.isnan(coll.sum(axis=1)) == False]   # noqa: E712
        out.append(np.sum(coll, axis=0) / len(gt_fix_vectors[index]) if coll.shape[0] != 0 else np.array([np.nan] * 2))
    return np.array(out)
