"""CPU: oracles of SURVEY.md §8 rows f1 / f2 / f4 against the reference's outputs and against first principles."""
import itertools
import json
import math
import os

import numpy as np

from helpers import GOLDEN
from oracle import sampling_oracle as SO


def test_collate_oracle_matches_reference_getitem_and_collate():
    """bit-exact with the reference's AiR.__getitem__ + collate_func (tests/golden/collate.npz, make_golden_collate.py): empty
    scanpaths, scanpaths longer than max_length, exact cell boundaries and the last pixel are among the records"""
    g = np.load(os.path.join(GOLDEN, "collate.npz"))
    recs = json.loads(bytes(g["records"]).decode())
    t, d, a, m = SO.collate_targets(recs, 16, (30, 40))
    assert np.array_equal(t, g["scanpaths"]) and np.array_equal(d, g["durations"])
    assert np.array_equal(a, g["action_masks"]) and np.array_equal(m, g["duration_masks"])
    assert (t.sum(-1) == 1).all()                       # every step has exactly one target (a cell or terminate)
    lens = [min(len(r["X"]), 16) for r in recs]
    assert [int(x) for x in m.sum(1)] == lens and [int(x) for x in a.sum(1)] == [min(n + 1, 16) for n in lens]


def _brute(probs, min_length, K):
    T, A = probs.shape
    best = {}
    for seq in itertools.product(range(A), repeat=T):
        s, out, ok = 0.0, [], True
        for t, a in enumerate(seq):
            if a == 0 and t < min_length:
                ok = False
                break
            if not probs[t, a] > 0:
                ok = False
                break
            s += math.log(float(probs[t, a]))
            out.append(a)
            if a == 0:
                break
        if ok:
            key = tuple(out + [0] * (T - len(out)))
            best[key] = s                                # the same terminated prefix always has the same score
    return sorted(best.items(), key=lambda kv: -kv[1])[:K]


def test_beam_oracle_is_exact_on_exhaustively_enumerable_cases():
    rng = np.random.Generator(np.random.PCG64(3))
    for T, A, K, ml in ((3, 5, 4, 1), (4, 4, 3, 2), (2, 6, 4, 0), (4, 3, 2, 1)):
        p = rng.random((T, A)).astype(np.float32) ** 3
        p[rng.integers(T), rng.integers(1, A)] = 0.0
        p /= p.sum(-1, keepdims=True)
        acts, scores = SO.beam_search(p, ml, K)
        want = _brute(p, ml, K)
        assert [tuple(a) for a in acts[:len(want)]] == [k for k, _ in want], (T, A, K, ml)
        assert np.allclose(scores[:len(want)], [v for _, v in want], rtol=0, atol=1e-12)
        assert (np.diff(scores[:len(want)]) <= 1e-15).all()


def test_beam_width_one_is_greedy_with_the_termination_rule():
    rng = np.random.Generator(np.random.PCG64(4))
    p = rng.random((16, 1201)).astype(np.float32)
    p[5, 0] = 10.0                                       # terminate becomes the most probable action at step 5
    p /= p.sum(-1, keepdims=True)
    acts, _ = SO.beam_search(p, 2, 1)
    greedy = [int(p[t, (1 if t < 2 else 0):].argmax()) + (1 if t < 2 else 0) for t in range(16)]
    assert list(acts[0][:6]) == greedy[:5] + [0] and (acts[0][6:] == 0).all()


def test_multimatch_restatement_first_principles():
    """restated algorithm (multimatch_gaze is absent: parity unpinned) -- analytic cases"""
    from scanpaths_amd.utils.evaltools.multimatch import docomparison
    dt = {"names": ("start_x", "start_y", "duration"), "formats": ("f8", "f8", "f8")}

    def fv(xy, d):
        a = np.zeros(len(xy), dtype=dt)
        a["start_x"], a["start_y"], a["duration"] = [p[0] for p in xy], [p[1] for p in xy], d
        return a
    a = fv([(10, 10), (110, 10), (110, 90), (200, 90)], [0.2, 0.3, 0.25, 0.4])
    assert docomparison(a, a, screensize=[320, 240]) == [1.0] * 5
    b = fv([(40, 50), (140, 50), (140, 130), (230, 130)], [0.2, 0.3, 0.25, 0.4])        # the same shape shifted by (30, 40)
    r = docomparison(a, b, screensize=[320, 240])
    assert r[0] == r[1] == r[2] == r[4] == 1.0 and abs(r[3] - (1 - 50.0 / 400.0)) < 1e-12
    c = fv([(10, 10), (110, 10), (110, 90), (200, 90)], [0.4, 0.6, 0.5, 0.8])            # durations doubled
    assert abs(docomparison(a, c, screensize=[320, 240])[4] - 0.5) < 1e-12
    assert all(math.isnan(v) for v in docomparison(a, fv([(1, 1), (2, 2)], [0.1, 0.1]), screensize=[320, 240]))
