"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Literal restatement of ``evaluation_performance_related``
(/root/reference/AiR/utils/evaluation.py:188-359): the reference's nested loops, one pair at a time, on the CPU oracles of
ScanMatch (oracle/scanmatch_oracle.py) and SED / STDE (oracle/metrics_oracle.py), both pinned bit-exact to the reference's own
known answers.  MultiMatch is the caller-supplied ``docomparison`` (the third-party package is absent: parity of that column is
unpinned, see scanpaths_amd/utils/evaltools/multimatch.py).

PINNED (round 3): tests/golden/eval_metrics.npz holds the outputs of the REAL reference functions (imported in the build container
by tests/golden/make_golden_eval.py with a deterministic ``multimatch_gaze`` stand-in, tests/helpers.py::toy_multimatch) for seeded
scanpath sets; tests/test_oracle_golden.py holds both functions below to them (pair enumeration and dropping, grouping, float32
collection, means / stds, the "best" columns, per-image scores)."""
import numpy as np

from . import metrics_oracle as MO
from . import scanmatch_oracle as SO


def evaluation_performance_related(gt_fix_vectors, predict_fix_vectors, all_performances, all_allocated_performances, docomparison):
    S = SO.submatrix(16, 12, 3.5)                                                                        # :198-199
    stimulus_shape = (240, 320, 3)
    collect_all, collect_right, collect_wrong, scores_of_each_images = [], [], [], []
    for index in range(len(gt_fix_vectors)):                                                              # :205
        gt_fix_vector, predict_fix_vector = gt_fix_vectors[index], predict_fix_vectors[index]
        performances = all_performances[index]
        sample_all, sample_right, sample_wrong = [], [], []
        for inner_index in range(len(gt_fix_vector)):
            inner = gt_fix_vector[inner_index]
            row = list(docomparison(inner, predict_fix_vector, screensize=[320, 240]))                    # :213
            if np.any(np.isnan(row)):
                continue
            f1 = np.array([list(_) for _ in list(inner)])
            f2 = np.array([list(_) for _ in list(predict_fix_vector)])
            f1[:, -1] *= 1000
            f2[:, -1] *= 1000
            for tempbin in (50.0, 0.0):                                                                   # with, then without duration
                s1 = SO.fixation_to_sequence(f1, 320, 240, 16, 12, (0, 0), tempbin).astype(np.int32)
                s2 = SO.fixation_to_sequence(f2, 320, 240, 16, 12, (0, 0), tempbin).astype(np.int32)
                row.append(SO.nw_score(s1, s2, S, 0.0))
            row.append(MO.sed(stimulus_shape, f1, f2))                                                    # :238
            row.append(MO.stde(f1, f2, stimulus_shape))                                                   # :242
            sample_all.append(row)
            if performances[inner_index] == True and all_allocated_performances[index] == True:           # noqa: E712
                sample_right.append(row)
            elif performances[inner_index] == False and all_allocated_performances[index] == False:       # noqa: E712
                sample_wrong.append(row)
        collect_all.append(np.array(sample_all, dtype=np.float32))
        collect_right.append(np.array(sample_right, dtype=np.float32))
        collect_wrong.append(np.array(sample_wrong, dtype=np.float32))
        chosen = sample_right if all_allocated_performances[index] == True else sample_wrong             # noqa: E712
        scores_of_each_images.append(list(np.array(chosen).mean(axis=0)) if chosen != [] else list(np.zeros((9,))))
    summary_mean, summary_std = [], []
    for coll in (collect_all, collect_right, collect_wrong):
        coll = [a for a in coll if len(a) != 0]
        rl = np.concatenate(coll, axis=0)
        tmp = np.concatenate([np.concatenate([[a[:, 7].min(keepdims=True), a[:, 8].max(keepdims=True)]]).transpose((1, 0))
                              for a in coll], axis=0)
        summary_mean.append(np.concatenate([rl.mean(0), tmp.mean(0)], axis=0))
        summary_std.append(np.concatenate([rl.std(0), tmp.std(0)], axis=0))
    return summary_mean, summary_std, scores_of_each_images


def _pair_row(fv1, fv2, docomparison, S, stimulus_shape):
    """one 9-column row of metrics for the ordered pair (fv1, fv2), or None when MultiMatch cannot score it (:45-47 / :215-217)"""
    row = list(docomparison(fv1, fv2, screensize=[320, 240]))
    if np.any(np.isnan(row)):
        return None
    f1 = np.array([list(_) for _ in list(fv1)])
    f2 = np.array([list(_) for _ in list(fv2)])
    f1[:, -1] *= 1000
    f2[:, -1] *= 1000
    for tempbin in (50.0, 0.0):
        s1 = SO.fixation_to_sequence(f1, 320, 240, 16, 12, (0, 0), tempbin).astype(np.int32)
        s2 = SO.fixation_to_sequence(f2, 320, 240, 16, 12, (0, 0), tempbin).astype(np.int32)
        row.append(SO.nw_score(s1, s2, S, 0.0))
    row.append(MO.sed(stimulus_shape, f1, f2))
    row.append(MO.stde(f1, f2, stimulus_shape))
    return row


def _summary(collects):
    summary_mean, summary_std = [], []
    for coll in collects:
        coll = [a for a in coll if len(a) != 0]                                                           # the `_ != []` idiom (:108-110)
        rl = np.concatenate(coll, axis=0)
        tmp = np.concatenate([np.concatenate([[a[:, 7].min(keepdims=True), a[:, 8].max(keepdims=True)]]).transpose((1, 0))
                              for a in coll], axis=0)
        summary_mean.append(np.concatenate([rl.mean(0), tmp.mean(0)], axis=0))
        summary_std.append(np.concatenate([rl.std(0), tmp.std(0)], axis=0))
    return summary_mean, summary_std


def human_evaluation(dataloader, docomparison):
    """(/root/reference/AiR/utils/evaluation.py:11-186) every ORDERED pair of distinct human scanpaths of an image.
    -> summary_mean [3][11], summary_std [3][11], {question_id: {True: good row, False: poor row}}"""
    S = SO.submatrix(16, 12, 3.5)
    stimulus_shape = (240, 320, 3)
    collect_all, collect_right, collect_wrong, good_scores, poor_scores, names = [], [], [], [], [], []
    for batch in dataloader:                                                                              # :27
        names.extend(batch["question_ids"])
        for fix_vectors, performances in zip(batch["fix_vectors"], batch["performances"]):
            sample_all, sample_right, sample_wrong = [], [], []
            for index_1 in range(len(fix_vectors)):
                for index_2 in range(len(fix_vectors)):
                    if index_2 == index_1:
                        continue
                    row = _pair_row(fix_vectors[index_1], fix_vectors[index_2], docomparison, S, stimulus_shape)
                    if row is None:
                        continue
                    sample_all.append(row)
                    if performances[index_1] == True and performances[index_2] == True:                    # noqa: E712  (:78)
                        sample_right.append(row)
                    elif performances[index_1] == False and performances[index_2] == False:                # noqa: E712  (:81)
                        sample_wrong.append(row)
            collect_all.append(np.array(sample_all, dtype=np.float32))
            collect_right.append(np.array(sample_right, dtype=np.float32))
            collect_wrong.append(np.array(sample_wrong, dtype=np.float32))
            good_scores.append(list(np.array(sample_right, dtype=np.float64).mean(axis=0)) if sample_right != [] else list(np.zeros((9,))))
            poor_scores.append(list(np.array(sample_wrong, dtype=np.float64).mean(axis=0)) if sample_wrong != [] else list(np.zeros((9,))))
    summary_mean, summary_std = _summary((collect_all, collect_right, collect_wrong))
    return summary_mean, summary_std, {n: {True: g, False: p} for n, g, p in zip(names, good_scores, poor_scores)}
