"""The ScanMatch oracle (oracle/scanmatch_oracle.py) against the outputs of the real reference stored in
tests/golden/scanmatch.npz (its own .mat example + seeded random scanpaths): float64, bit-exact."""
import os

import numpy as np
import pytest

from oracle import scanmatch_oracle as SO

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "scanmatch.npz"))


def unrag(name):
    cat, off = GOLD[name], GOLD[name + "_off"]
    return [cat[off[i]:off[i + 1]] for i in range(len(off) - 1)]


def fixes(name):
    cat, off = GOLD[name], GOLD[name + "_off"]
    return [cat[3 * off[i]:3 * off[i + 1]].reshape(-1, 3) for i in range(len(off) - 1)]


def same(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.array_equal(a, b, equal_nan=True), np.nanmax(np.abs(a - b))


def test_known_answers_of_the_reference_self_check():
    """the three scores of the reference's __main__ block with and without duration"""
    wd, wod = GOLD["ex_scores_wd"], GOLD["ex_scores_wod"]
    assert (wd[0, 1], wd[0, 2], wd[1, 2]) == (0.6725138474550876, 0.22829669183275586, 0.253819062877192)
    assert (wod[0, 1], wod[0, 2], wod[2, 2]) == (0.6178313750019084, 0.2582431346483109, 1.0)


def test_submatrix_and_grid_bit_exact():
    same(SO.submatrix(12, 8, 3.5), GOLD["ex_submatrix"])
    xs = SO.bin_of_pixel(np.arange(1024), 12, 1024)
    ys = SO.bin_of_pixel(np.arange(768), 8, 768)
    same(ys[0] * 12 + xs, GOLD["ex_mask_row0"])
    same(ys * 12 + xs[0], GOLD["ex_mask_col0"])


@pytest.mark.parametrize("tag,cfg", [
    ("ex_seq_wd", dict(Xres=1024, Yres=768, Xbin=12, Ybin=8, tempbin=100.0)),
    ("ex_seq_wod", dict(Xres=1024, Yres=768, Xbin=12, Ybin=8, tempbin=0.0)),
    ("rnd_seq_wd", dict(Xres=320, Yres=240, Xbin=16, Ybin=12, tempbin=50.0)),
    ("rnd_seq_wod", dict(Xres=320, Yres=240, Xbin=16, Ybin=12, tempbin=0.0)),
    ("gap_seq", dict(Xres=300, Yres=200, Xbin=10, Ybin=7, tempbin=80.0, offset=(10, 20))),
])
def test_sequences_bit_exact(tag, cfg):
    fx = fixes("ex_fix" if tag.startswith("ex") else "rnd_fix")
    want = unrag(tag)
    for f, w in zip(fx, want):
        same(SO.fixation_to_sequence(f, **cfg), w)
    if tag == "rnd_seq_wd":
        assert len(want[5]) == 0            # all durations round to zero repeats
        assert len(want[6]) == 0            # 25/50 = 0.5 -> round-half-even -> 0
        assert len(want[7]) == 2 * len(fx[7])   # 75/50 = 1.5 -> 2


@pytest.mark.parametrize("seqs,scores,cfg", [
    ("ex_seq_wd", "ex_scores_wd", (12, 8, 3.5, 0.0)),
    ("ex_seq_wod", "ex_scores_wod", (12, 8, 3.5, 0.0)),
    ("rnd_seq_wod", "rnd_scores_wod", (16, 12, 3.5, 0.0)),
    ("gap_seq", "gap_scores", (10, 7, 2.0, -0.75)),
])
def test_scores_bit_exact(seqs, scores, cfg):
    S = SO.submatrix(*cfg[:3])
    sq = unrag(seqs)
    want = GOLD[scores]
    n = min(len(sq), 12)                    # pure-python DP: keep the CPU suite short
    got = np.array([[SO.nw_score(sq[i], sq[j], S, cfg[3]) for j in range(n)] for i in range(n)])
    same(got, want[:n, :n])


def test_empty_sequences_follow_the_reference():
    S = SO.submatrix(16, 12, 3.5)
    sq = unrag("rnd_seq_wd")
    want = GOLD["rnd_scores_wd"]
    for i, j in [(5, 5), (5, 0), (0, 5), (6, 7), (7, 7), (5, 6)]:
        got = SO.nw_score(sq[i], sq[j], S, 0.0)
        assert (np.isnan(got) and np.isnan(want[i, j])) or got == want[i, j], (i, j, got, want[i, j])
    assert np.isnan(want[5, 5]) and want[5, 0] == 0.0


def test_alignment_and_F_bit_exact():
    S = SO.submatrix(12, 8, 3.5)
    a, b = unrag("ex_seq_wd")[:2]
    score, align, F = SO.nw_match(a, b, S, 0.0)
    assert score == float(GOLD["ex_match01_score"])
    same(align, GOLD["ex_match01_align"])
    same(F, GOLD["ex_match01_F"])
    b2, c2 = unrag("ex_seq_wod")[1:3]
    score, align, F = SO.nw_match(b2, c2, S, 0.0)
    assert score == float(GOLD["ex_match12_wod_score"])
    same(align, GOLD["ex_match12_wod_align"])
    same(F, GOLD["ex_match12_wod_F"])
    Sg = SO.submatrix(10, 7, 2.0)
    g0, g1 = unrag("gap_seq")[:2]
    _, align, F = SO.nw_match(g0, g1, Sg, -0.75)
    same(align, GOLD["gap_match01_align"])
    same(F, GOLD["gap_match01_F"])


# ---- SED / STDE ------------------------------------------------------------------------------------------------
from oracle import metrics_oracle as MO   # noqa: E402

GOLD2 = np.load(os.path.join(os.path.dirname(__file__), "golden", "sed_stde.npz"))


def _rnd_fixes():
    cat, off = GOLD2["rnd_fix"], GOLD2["rnd_fix_off"]
    return [cat[off[i]:off[i + 1]] for i in range(len(off) - 1)]


def test_sed_stde_known_answers_of_the_reference_self_check():
    ex = fixes("ex_fix")
    got_sed = np.array([[MO.sed((768, 1024, 3), a, b) for b in ex] for a in ex])
    assert np.array_equal(got_sed, GOLD2["ex_sed"]) and got_sed[0, 1] == 9
    got = np.array([[MO.stde(a, b, (768, 1024, 3)) for b in ex] for a in ex])
    same(got, GOLD2["ex_stde"])
    assert got[0, 1] == 0.9064806433533912


def test_sed_stde_random_pairs_bit_exact():
    fx = _rnd_fixes()
    n = 14
    got_sed = np.array([[MO.sed((240, 320, 3), fx[i], fx[j]) for j in range(n)] for i in range(n)])
    assert np.array_equal(got_sed, GOLD2["rnd_sed"][:n, :n])
    got8 = np.array([[MO.sed((240, 320, 3), fx[i], fx[j], n=8) for j in range(12)] for i in range(12)])
    assert np.array_equal(got8, GOLD2["rnd_sed_n8"])
    got = np.array([[MO.stde(fx[i], fx[j], (240, 320, 3)) for j in range(n)] for i in range(n)])
    same(got, GOLD2["rnd_stde"][:n, :n])


# ---- RL reward glue: host grouping logic against the reference, with an oracle-backed scorer -----------------
GOLD3 = np.load(os.path.join(os.path.dirname(__file__), "golden", "rl.npz"))
_DT = {"names": ("start_x", "start_y", "duration"), "formats": ("f8", "f8", "f8")}


class OracleScanMatch:
    """the reference's ScanMatch call surface on top of oracle/scanmatch_oracle.py (tests only)"""
    def __init__(self, tempbin):
        self.tempbin = tempbin
        self.S = SO.submatrix(16, 12, 3.5)

    def fixationToSequence(self, data):
        return SO.fixation_to_sequence(data, 320, 240, 16, 12, tempbin=self.tempbin).astype(np.float64)

    def match(self, a, b):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return SO.nw_score(a, b, self.S, 0.0), None, None


def rl_case():
    def unflat(fix, lens):
        out, o = [], 0
        for n in lens:
            fv = np.zeros(int(n), dtype=_DT)
            fv["start_x"], fv["start_y"], fv["duration"] = fix[o:o + n, 0], fix[o:o + n, 1], fix[o:o + n, 2]
            out.append(fv)
            o += int(n)
        return out
    flat = unflat(GOLD3["gt_fix"], GOLD3["gt_len"])
    gt, perf, o = [], [], 0
    for c in GOLD3["gt_count"]:
        gt.append(flat[o:o + int(c)])
        perf.append([bool(v) for v in GOLD3["perf"][o:o + int(c)]])
        o += int(c)
    return gt, perf, unflat(GOLD3["pred_fix"], GOLD3["pred_len"])


def check_rl_glue(wd, wod):
    from scanpaths_amd.utils.evaluation import (gtpairs_eval_scanmatch_performance_related,
                                                pairs_eval_scanmatch_performance_related)
    gt, perf, pred = rl_case()
    for given, tag in ((True, "good"), (False, "poor")):
        s, d, acc = pairs_eval_scanmatch_performance_related(gt, pred, wd, wod, perf, given)
        same(s, GOLD3[f"pairs_{tag}_same"])
        same(d, GOLD3[f"pairs_{tag}_diff"])
        assert int(acc) == int(GOLD3[f"pairs_{tag}_accept"])
    g, p, dd = gtpairs_eval_scanmatch_performance_related(gt, wd, wod, perf)
    same(g, GOLD3["gtpairs_good"])
    same(p, GOLD3["gtpairs_poor"])
    same(dd, GOLD3["gtpairs_diff"])


def test_rl_reward_glue_matches_reference():
    check_rl_glue(OracleScanMatch(50.0), OracleScanMatch(0.0))
