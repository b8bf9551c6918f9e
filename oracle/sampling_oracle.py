"""ORACLE -- TEST INFRASTRUCTURE ONLY (imported by tests/ only; the product never does).

CPU restatements for SURVEY.md §8 rows f1 / f4:
  * ``collate_targets``  -- the target construction of AiR/dataset/dataset.py:111-147 (blur_sigma = None) in plain numpy, line
                            by line.  PARITY: pinned -- tests/golden/collate.npz holds outputs of the reference's own
                            ``AiR.__getitem__`` + ``collate_func`` on synthetic files (tests/golden/make_golden_collate.py).
  * ``beam_search``      -- build-side decoder (the reference only samples, models/sampling.py:16-46): exhaustive-by-
                            construction width-K beam over independent per-step distributions.  No reference exists for it:
                            "parity unpinned"; the tests check HIP == this restatement and the optimality property directly.
"""
from __future__ import annotations

import math

import numpy as np


def collate_targets(fixations, max_length, action_map, f64_div=False, blur_sigma=None):
    """fixations: list of dicts with X, Y, T_start, T_end (lists), height, width.  Returns target_scanpath [B,T,1+h*w],
    duration, action_mask, duration_mask [B,T] float32 -- AiR/dataset/dataset.py:111-147.  blur_sigma: :144-146 with scipy's own
    gaussian_filter, as the reference calls it.  f64_div: numpy 1.x semantics (PARITY: tests/golden/collate_f64.npz holds the real
    reference's outputs under float64 division, see tests/golden/make_golden_dataset.py)."""
    import scipy.ndimage as filters
    H, W = action_map
    outs = [[], [], [], []]
    for fixation in fixations:
        downscale_x = fixation["width"] / W                                           # :109
        downscale_y = fixation["height"] / H                                          # :110
        scanpath = np.zeros((max_length, H, W), dtype=np.float32)
        target_scanpath = np.zeros((max_length, H * W + 1), dtype=np.float32)         # :114
        duration = np.zeros(max_length, dtype=np.float32)
        action_mask = np.zeros(max_length, dtype=np.float32)
        duration_mask = np.zeros(max_length, dtype=np.float32)
        pos_x = np.array(fixation["X"]).astype(np.float32)                            # :119
        pos_y = np.array(fixation["Y"]).astype(np.float32)
        duration_raw = np.array(fixation["T_end"]).astype(np.float32) - np.array(fixation["T_start"]).astype(np.float32)
        pos_x_discrete = np.zeros(max_length, dtype=np.int32) - 1
        pos_y_discrete = np.zeros(max_length, dtype=np.int32) - 1
        for index in range(len(pos_x)):                                               # :125-134
            if index == max_length:
                break
            if f64_div:      # numpy 1.x value-based casting: float32 scalar / python float -> float64
                pos_x_discrete[index] = np.int32(np.float64(pos_x[index]) / downscale_x)
                pos_y_discrete[index] = np.int32(np.float64(pos_y[index]) / downscale_y)
            else:            # numpy >= 2 (NEP 50): the python float is weak -> float32 division
                pos_x_discrete[index] = (pos_x[index] / np.float32(downscale_x)).astype(np.int32)
                pos_y_discrete[index] = (pos_y[index] / np.float32(downscale_y)).astype(np.int32)
            duration[index] = np.float64(duration_raw[index]) / 1000.0 if f64_div else duration_raw[index] / np.float32(1000.0)
            action_mask[index] = 1
            duration_mask[index] = 1
        if action_mask.sum() <= max_length - 1:                                       # :135-136
            action_mask[int(action_mask.sum())] = 1
        for index in range(max_length):                                               # :139-147
            if pos_x_discrete[index] == -1 or pos_y_discrete[index] == -1:
                target_scanpath[index, 0] = 1
            else:
                scanpath[index, pos_y_discrete[index], pos_x_discrete[index]] = 1
                if blur_sigma:                                                        # :144-146
                    scanpath[index] = filters.gaussian_filter(scanpath[index], blur_sigma)
                    scanpath[index] /= scanpath[index].sum()
                target_scanpath[index, 1:] = scanpath[index].reshape(-1)
        for o, v in zip(outs, (target_scanpath, duration, action_mask, duration_mask)):
            o.append(v)
    return tuple(np.stack(o) for o in outs)


def beam_search(probs, min_length, K):
    """probs [T, A] float32 of ONE sample -> (actions [K, T] int64, scores [K] float64): the K best sequences under
    sum_t log p_t(a_t), terminate (action 0) ends a sequence and is allowed from t >= min_length.  Same tie-breaking as
    csrc/sampling.hip beam_kernel: per step the K most probable allowed actions (probability desc, index asc); candidates ordered
    (beam, action rank); stable selection of the K highest scores."""
    T, A = probs.shape
    beams = [(0.0, [], False)]
    for t in range(T):
        p = probs[t]
        lo = 1 if t < min_length else 0
        idx = sorted(range(lo, A), key=lambda a: (-float(p[a]), a))[:K]
        cand = []
        for (s, seq, done) in beams:
            if done:
                cand.append((s, seq + [0], True))
                continue
            for a in idx:
                if not p[a] > 0:
                    continue
                cand.append((s + math.log(float(p[a])), seq + [a], a == 0))
        order = sorted(range(len(cand)), key=lambda c: (-cand[c][0], c))[:K]
        beams = [cand[c] for c in order]
    acts = np.zeros((K, T), dtype=np.int64)
    scores = np.full(K, -np.inf)
    for q, (s, seq, _) in enumerate(beams):
        acts[q] = seq
        scores[q] = s
    return acts, scores
