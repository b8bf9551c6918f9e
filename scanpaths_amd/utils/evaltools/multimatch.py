"""MultiMatch (Dewhurst et al. 2012, Jarodzka et al. 2010) -- the five scanpath similarities the reference obtains from the
third-party ``multimatch_gaze.docomparison`` (utils/evaluation.py:8,43,213; multimatch_gaze==0.1.2, sp_baseline.yml:65).

That package is NOT vendored in the reference and is absent here, so this is a restatement of its published algorithm with the
reference's call arguments (screensize=[320, 240], no grouping / simplification): saccade vectors of both scanpaths -> matrix of
vector differences -> cheapest monotone alignment path from the first to the last saccade pair (steps right / down / diagonal,
cost = the entered cell) -> medians of the vector, direction, length, position and duration differences along the path ->
normalisation to [0, 1].  Scanpaths with fewer than 3 fixations give five NaNs (the rule that makes the reference drop a pair).
PARITY: **unpinned** -- no fixture of the original package exists in the reference; utils/evaluation.py uses the installed
``multimatch_gaze`` instead whenever it is importable.

Two forms: ``docomparison`` (host numpy, one pair; the reference's call signature) and ``multimatch_pairs`` (round 4: ALL pairs of a
validation call in one launch of ``sp_scan_multimatch``, one thread per pair -- the triple python loop per pair was the wall-clock of
validation once ScanMatch / SED / STDE ran on the device).  The host form is the device kernel's checker (tests/test_scanmatch_gpu.py)."""
from __future__ import annotations

import math

import numpy as np


def _structure(data):
    a = np.array([list(_) for _ in list(data)], dtype=np.float64).reshape(-1, 3)
    x, y, dur = a[:, 0], a[:, 1], a[:, 2]
    lenx, leny = x[1:] - x[:-1], y[1:] - y[:-1]
    return {"fx": x, "fy": y, "dur": dur, "sx": x[:-1], "sy": y[:-1], "lenx": lenx, "leny": leny,
            "rho": np.sqrt(lenx ** 2 + leny ** 2), "theta": np.arctan2(leny, lenx)}


def _alignment(M):
    """cheapest path (0,0) -> (n-1,m-1) over moves right / down / diagonal, cost of a move = M at the cell entered"""
    n, m = M.shape
    D = np.full((n, m), np.inf)
    prev = np.zeros((n, m, 2), dtype=np.int64)
    D[0, 0] = 0.0
    for i in range(n):
        for j in range(m):
            if i == 0 and j == 0:
                continue
            best, arg = np.inf, (0, 0)
            for di, dj in ((0, 1), (1, 0), (1, 1)):
                pi, pj = i - di, j - dj
                if pi >= 0 and pj >= 0 and D[pi, pj] + M[i, j] < best:
                    best, arg = D[pi, pj] + M[i, j], (pi, pj)
            D[i, j] = best
            prev[i, j] = arg
    path = [(n - 1, m - 1)]
    while path[-1] != (0, 0):
        i, j = path[-1]
        path.append((int(prev[i, j, 0]), int(prev[i, j, 1])))
    return path[::-1]


def docomparison(fixation_vectors1, fixation_vectors2, screensize, grouping=False, TDir=0.0, TDur=0.0, TAmp=0.0):
    if grouping:
        raise NotImplementedError("scanpath simplification (grouping=True) is not used by the reference and not restated")
    if not (len(fixation_vectors1) >= 3 and len(fixation_vectors2) >= 3):
        return [np.nan] * 5
    p1, p2 = _structure(fixation_vectors1), _structure(fixation_vectors2)
    M = np.sqrt((p1["lenx"][:, None] - p2["lenx"][None, :]) ** 2 + (p1["leny"][:, None] - p2["leny"][None, :]) ** 2)
    path = _alignment(M)
    vec, ang, ln, pos, dur = [], [], [], [], []
    for i, j in path:
        vec.append(math.sqrt((p1["lenx"][i] - p2["lenx"][j]) ** 2 + (p1["leny"][i] - p2["leny"][j]) ** 2))
        t = [p1["theta"][i], p2["theta"][j]]
        t = [math.pi + (math.pi + v) if v < 0 else v for v in t]
        d = abs(t[0] - t[1])
        ang.append(2 * math.pi - d if d > math.pi else d)
        ln.append(abs(p1["rho"][i] - p2["rho"][j]))
        pos.append(math.sqrt((p1["sx"][i] - p2["sx"][j]) ** 2 + (p1["sy"][i] - p2["sy"][j]) ** 2))
        dur.append(abs(p1["dur"][i] - p2["dur"][j]) / max(p1["dur"][i], p2["dur"][j]))
    un = [float(np.median(v)) for v in (vec, ang, ln, pos, dur)]
    diag = math.sqrt(screensize[0] ** 2 + screensize[1] ** 2)
    return [1 - un[0] / (2 * diag), 1 - un[1] / math.pi, 1 - un[2] / diag, 1 - un[3] / diag, 1 - un[4]]


def multimatch_pairs(scanpaths, pairs, screensize):
    """MultiMatch of many pairs on the device.  scanpaths: list of fixation records / arrays (x, y, duration); pairs: [npairs, 2]
    indices into scanpaths (first, second argument of docomparison); screensize [width, height].  Returns float64 [npairs, 5]
    (numpy), five NaNs for a pair with a scanpath of fewer than 3 fixations."""
    import torch

    from ... import hip
    from ...hip import check, ptr
    pr = np.asarray(pairs, dtype=np.int32).reshape(-1, 2)
    if pr.shape[0] == 0:
        return np.zeros((0, 5))
    L = hip.lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    arrs = [np.array([list(_) for _ in list(a)], dtype=np.float64).reshape(-1, 3) for a in scanpaths]
    counts = [a.shape[0] for a in arrs]
    if max(counts) > L.sp_scan_max_fixations():
        raise ValueError(f"scanpath of {max(counts)} fixations exceeds the kernel limit {L.sp_scan_max_fixations()}")
    count = torch.tensor(counts, dtype=torch.int32)
    start = (torch.cumsum(count.to(torch.int64), 0) - count.to(torch.int64)).to(dev)
    cat = np.concatenate(arrs, 0) if sum(counts) else np.zeros((1, 3))
    fix = torch.from_numpy(np.ascontiguousarray(cat)).to(dev)
    prd = torch.from_numpy(np.ascontiguousarray(pr)).to(dev)
    count_d = count.to(dev)
    out = torch.empty((pr.shape[0], 5), dtype=torch.float64, device=dev)
    check(L.sp_scan_multimatch(ptr(fix), 3, ptr(start), ptr(count_d), ptr(prd), pr.shape[0], float(screensize[0]), float(screensize[1]),
                               ptr(out), hip.stream()), "sp_scan_multimatch")
    return out.cpu().numpy()
