"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy, float64 / int) of the two string / embedding scanpath metrics the
reference's evaluation computes next to ScanMatch.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.

Restated from the behaviour of /root/reference/AiR/utils/evaltools/visual_attention_metrics.py:
  sed()   string_edit_distance :300-318 -> _scanpath_to_string :288-299 (cell = int32(x) // (width // n) + int32(y) // (height // n) * n)
          -> _Levenshtein :236-285 (unit insert / delete / substitute costs, returns the integer distance)
  stde()  scaled_time_delay_embedding_similarity :392-441 (coordinates / max(image.shape); for every window length
          k = 1..min(len): time_delay_embedding_distance :332-389 in 'Mean' mode = mean over simulated k-windows of the minimum
          over human k-windows of sum_i ||s_i - h_i|| / k (euclidean_distance :205-218); similarity = mean_k exp(-distance_k))
Call order in the reference's evaluation (utils/evaluation.py:68-72): sed(stimulus, fix_1, fix_2), stde(fix_1, fix_2, stimulus).
Parity pinned: tests/test_scanmatch_oracle.py compares both against tests/golden/sed_stde.npz (outputs of the reference on
its own .mat example and on 40 seeded random scanpaths, tests/golden/make_golden_sed_stde.py) -- bit-exact."""
from __future__ import annotations

import numpy as np


def grid_string(fix: np.ndarray, height: int, width: int, n: int = 5) -> np.ndarray:
    f = np.asarray(fix)[:, :2].astype(np.int32)
    return f[:, 0] // (width // n) + f[:, 1] // (height // n) * n


def levenshtein(a, b) -> int:
    prev = list(range(len(b) + 1))
    for i in range(1, len(a) + 1):
        cur = [i] + [0] * len(b)
        for j in range(1, len(b) + 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (a[i - 1] != b[j - 1]))
        prev = cur
    return int(prev[len(b)])


def sed(stimulus_shape, human: np.ndarray, simulated: np.ndarray, n: int = 5) -> int:
    h, w = stimulus_shape[0], stimulus_shape[1]
    return levenshtein(grid_string(human, h, w, n), grid_string(simulated, h, w, n))


def stde(human: np.ndarray, simulated: np.ndarray, image_shape) -> float:
    md = float(max(image_shape))
    H = np.asarray(human, dtype=np.float64)[:, :2] / md
    S = np.asarray(simulated, dtype=np.float64)[:, :2] / md
    kmax = min(len(H), len(S))
    if kmax == 0:
        return float("nan")              # the reference returns None here
    sims = []
    for k in range(1, kmax + 1):
        dists = []
        for s0 in range(len(S) - k + 1):
            best = None
            for h0 in range(len(H) - k + 1):
                # argument order of the reference: euclidean_distance(s_k_vec, h_k_vec) -> (s - h) components
                d = np.sqrt((S[s0:s0 + k, 0] - H[h0:h0 + k, 0]) ** 2 + (S[s0:s0 + k, 1] - H[h0:h0 + k, 1]) ** 2).sum()
                best = d if best is None or d < best else best
            dists.append(best / k)
        sims.append(np.exp(-(sum(dists) / len(dists))))
    return float(sum(sims) / len(sims))
