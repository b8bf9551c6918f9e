"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy, float64) of the ScanMatch scorer the reference's evaluation and RL
reward use.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product path (scanpaths_amd/utils/evaltools/scanmatch.py -> csrc/scanmatch.hip) never does.

Algorithm: Cristino, Mathot, Theeuwes & Gilchrist (2010), "ScanMatch: a novel method for comparing fixation sequences",
Behav Res Methods 42(3) -- grid-binned fixation strings (optionally repeated per temporal bin), substitution matrix from
inter-bin Euclidean distance, Needleman-Wunsch global alignment, score normalised by max(substitution) * max(len).
Restated from the behaviour of the reference's implementation, citing /root/reference/AiR/utils/evaltools/scanmatch.py:
  submatrix()             CreateSubMatrix :88-103   (value order: |dist - max| - (max - threshold))
  bin_of_pixel()          GridMask        :105-115  (index = int32(pixel * (bins / res)), symbol = ybin * Xbin + xbin)
  fixation_to_sequence()  fixationToSequence :117-135 (offset, clamp, int truncation of ALL columns incl. duration,
                                                      repeats = round_half_even(duration / TempBin))
  nw_match()              match           :137-197  (border F[i,0] = gap*(i+1), F[0,j] = gap*(j+1); score = max(F) / (max(S)*max(n,m));
                                                      traceback preference diagonal > left(delete) > up(insert))
Parity pinned: tests/test_scanmatch_oracle.py checks every function against tests/golden/scanmatch.npz, which holds the
outputs of the reference itself on its own example fixture (the .mat known answers of SURVEY.md §4) and on seeded random
scanpaths (tests/golden/make_golden_scanmatch.py)."""
from __future__ import annotations

import numpy as np


def submatrix(Xbin: int, Ybin: int, threshold: float) -> np.ndarray:
    ys, xs = np.divmod(np.arange(Xbin * Ybin), Xbin)
    dist = np.sqrt(((xs[:, None] - xs[None, :]) ** 2 + (ys[:, None] - ys[None, :]) ** 2).astype(np.float64))
    mx = dist.max()
    return np.abs(dist - mx) - (mx - threshold)


def bin_of_pixel(px: np.ndarray, nbins: int, res: int) -> np.ndarray:
    return (np.asarray(px, dtype=np.float64) * (float(nbins) / res)).astype(np.int32)


def fixation_to_sequence(fix: np.ndarray, Xres: int, Yres: int, Xbin: int, Ybin: int, offset=(0, 0), tempbin: float = 0.0,
                         mask: np.ndarray | None = None) -> np.ndarray:
    d = np.array(fix, dtype=np.float64, copy=True)
    d[:, 0] -= offset[0]
    d[:, 1] -= offset[1]
    d[d < 0] = 0
    d[:, 0] = np.where(d[:, 0] >= Xres, Xres - 1, d[:, 0])
    d[:, 1] = np.where(d[:, 1] >= Yres, Yres - 1, d[:, 1])
    d = np.trunc(d).astype(np.int64)
    if mask is None:
        sym = bin_of_pixel(d[:, 1], Ybin, Yres).astype(np.int64) * Xbin + bin_of_pixel(d[:, 0], Xbin, Xres)
    else:
        sym = np.asarray(mask)[d[:, 1], d[:, 0]].astype(np.int64)
    if tempbin != 0:
        reps = np.round(d[:, 2] / float(tempbin)).astype(np.int64)
        sym = np.repeat(sym, reps)
    return sym.astype(np.int32)


def nw_fill(A, B, S: np.ndarray, gap: float) -> np.ndarray:
    n, m = len(A), len(B)
    F = np.zeros((n + 1, m + 1), dtype=np.float64)
    F[:, 0] = gap * (np.arange(n + 1) + 1)
    F[0, :] = gap * (np.arange(m + 1) + 1)
    for i in range(1, n + 1):
        srow = S[A[i - 1]]
        for j in range(1, m + 1):
            F[i, j] = max(F[i - 1, j - 1] + srow[B[j - 1]], F[i, j - 1] + gap, F[i - 1, j] + gap)
    return F


def nw_score(A, B, S: np.ndarray, gap: float = 0.0) -> float:
    F = nw_fill(A, B, S, gap)
    with np.errstate(invalid="ignore", divide="ignore"):
        return float(np.float64(F.max()) / np.float64(S.max() * max(len(A), len(B))))


def nw_match(A, B, S: np.ndarray, gap: float = 0.0):
    """(score, align [steps, 2] with -1 for a gap, F transposed [(m+1), (n+1)]) exactly as the reference returns them"""
    n, m = len(A), len(B)
    F = nw_fill(A, B, S, gap)
    al = []
    i, j = n, m
    while i > 0 and j > 0:
        if F[i, j] == F[i - 1, j - 1] + S[A[i - 1], B[j - 1]]:
            al.append((A[i - 1], B[j - 1])); i -= 1; j -= 1
        elif F[i, j] == F[i - 1, j] + gap:
            al.append((A[i - 1], -1)); i -= 1
        else:
            al.append((-1, B[j - 1])); j -= 1
    while i > 0:
        al.append((A[i - 1], -1)); i -= 1
    while j > 0:
        al.append((-1, B[j - 1])); j -= 1
    align = np.array(al[::-1], dtype=np.float64).reshape(-1, 2)
    with np.errstate(invalid="ignore", divide="ignore"):
        score = float(np.float64(F.max()) / np.float64(S.max() * max(n, m)))
    return score, align, F.T.copy()
