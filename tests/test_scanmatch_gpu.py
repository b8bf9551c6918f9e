"""HIP ScanMatch (csrc/scanmatch.hip through scanpaths_amd.utils.evaltools.scanmatch) against the reference's outputs in
tests/golden/scanmatch.npz and against the oracle on longer strings -- float64, BIT-EXACT (integer / IEEE max-plus work)."""
import os

import numpy as np
import pytest
import torch

from oracle import scanmatch_oracle as SO

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "scanmatch.npz"))


def unrag(name):
    cat, off = GOLD[name], GOLD[name + "_off"]
    return [cat[off[i]:off[i + 1]] for i in range(len(off) - 1)]


def fixes(name):
    cat, off = GOLD[name], GOLD[name + "_off"]
    return [cat[3 * off[i]:3 * off[i + 1]].reshape(-1, 3) for i in range(len(off) - 1)]


def same(a, b, what=""):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.array_equal(a, b, equal_nan=True), (what, np.nanmax(np.abs(a - b)))


def _sm(**kw):
    from scanpaths_amd.utils.evaltools.scanmatch import ScanMatch
    return ScanMatch(**kw)


def test_reference_self_check_known_answers():
    """the reference's __main__ block (scanmatch.py:222-257) through the drop-in API, incl. F and the alignment"""
    ex = fixes("ex_fix")
    cfg = dict(Xres=1024, Yres=768, Xbin=12, Ybin=8, Offset=(0, 0), Threshold=3.5)
    wd, wod = _sm(TempBin=100, **cfg), _sm(**cfg)
    same(wd.SubMatrix, GOLD["ex_submatrix"], "SubMatrix")
    m = wd.mask
    same(m[0], GOLD["ex_mask_row0"], "mask row")
    same(m[:, 0], GOLD["ex_mask_col0"], "mask col")
    s_wd = [wd.fixationToSequence(e).astype(np.int32) for e in ex]
    s_wod = [wod.fixationToSequence(e[:, :2]).astype(np.int32) for e in ex]
    for got, want in zip(s_wd, unrag("ex_seq_wd")):
        same(got, want, "seq wd")
    for got, want in zip(s_wod, unrag("ex_seq_wod")):
        same(got, want, "seq wod")
    s1, a1, f1 = wd.match(s_wd[0], s_wd[1])
    assert s1 == 0.6725138474550876 == float(GOLD["ex_match01_score"])
    same(a1, GOLD["ex_match01_align"], "align")
    same(f1, GOLD["ex_match01_F"], "F")
    assert wd.match(s_wd[0], s_wd[2])[0] == 0.22829669183275586
    assert wd.match(s_wd[1], s_wd[2])[0] == 0.253819062877192
    assert wod.match(s_wod[0], s_wod[1])[0] == 0.6178313750019084
    assert wod.match(s_wod[0], s_wod[2])[0] == 0.2582431346483109
    assert wod.match(s_wod[2], s_wod[2])[0] == 1.0
    s2, a2, f2 = wod.match(s_wod[1], s_wod[2])
    assert s2 == float(GOLD["ex_match12_wod_score"])
    same(a2, GOLD["ex_match12_wod_align"], "align wod")
    same(f2, GOLD["ex_match12_wod_F"], "F wod")
    with pytest.raises(ValueError, match="Unknown parameter"):
        _sm(Foo=1)


@pytest.mark.parametrize("seqs,scores,kw", [
    ("rnd_seq_wd", "rnd_scores_wd", dict(Xres=320, Yres=240, Xbin=16, Ybin=12, Offset=(0, 0), TempBin=50, Threshold=3.5)),
    ("rnd_seq_wod", "rnd_scores_wod", dict(Xres=320, Yres=240, Xbin=16, Ybin=12, Offset=(0, 0), Threshold=3.5)),
    ("gap_seq", "gap_scores", dict(Xres=300, Yres=200, Xbin=10, Ybin=7, Offset=(10, 20), Threshold=2.0, GapValue=-0.75,
                                   TempBin=80)),
])
def test_all_pairs_bit_exact_with_the_reference(seqs, scores, kw):
    """28 random scanpaths (out-of-screen fixations, empty strings after temporal binning, round-half-even durations,
    strings up to ~250 symbols = 4 column strips): sequences and the full 28x28 score matrix incl. the NaN of empty-vs-empty"""
    sm = _sm(**kw)
    fx = fixes("rnd_fix")
    sq, ln = sm.sequences(fx)
    want = unrag(seqs)
    assert ln.cpu().tolist() == [len(w) for w in want]
    for k, w in enumerate(want):
        same(sq[k, :len(w)].cpu().numpy(), w, f"sequence {k}")
    got = sm.match_all(fx, fx)
    same(got, GOLD[scores], scores)
    if seqs == "rnd_seq_wd":
        assert np.isnan(got[5, 5]) and got[5, 0] == 0.0
    # the single-pair drop-in agrees with the batched kernel
    assert sm.match(want[0], want[1])[0] == got[0, 1]


def test_long_strings_against_the_oracle():
    """strings of 1..700 symbols (up to 11 strips, partial last strips, 1-column last strip), both gap signs"""
    g = np.random.Generator(np.random.PCG64(7))
    lens = [1, 2, 63, 64, 65, 127, 128, 129, 193, 700, 5, 321]
    for gap in (0.0, -0.4, 0.3):
        sm = _sm(Xres=320, Yres=240, Xbin=16, Ybin=12, Threshold=3.5, GapValue=gap)
        S = SO.submatrix(16, 12, 3.5)
        same(sm.SubMatrix, S, "S")
        seqs = [g.integers(0, 192, L).astype(np.int32) for L in lens]
        ld = max(lens)
        dev = torch.device("cuda:0")
        buf = torch.zeros((len(seqs), ld), dtype=torch.int32)
        for k, s in enumerate(seqs):
            buf[k, :len(s)] = torch.from_numpy(s)
        buf = buf.to(dev)
        ln = torch.tensor(lens, dtype=torch.int32, device=dev)
        pairs = torch.tensor([(i, j) for i in range(len(lens)) for j in range(len(lens)) if (i + j) % 3 == 0 or i == 9 or j == 9],
                             dtype=torch.int32)
        pairs = pairs[:60]
        got = sm.match_pairs(buf, ln, buf, ln, pairs).cpu().numpy()
        for (i, j), v in zip(pairs.tolist(), got):
            assert v == SO.nw_score(seqs[i], seqs[j], S, gap), (gap, lens[i], lens[j])


def test_sequences_beyond_the_lds_limit_use_the_global_scratch_kernel():
    """A heavy-tailed sampled duration (RL phase) maps to more than sp_scanmatch_max_len() symbols: the scorer switches to the
    global-scratch kernel (same arithmetic: bit-exact with the oracle) instead of raising; beyond MAX_SYMBOLS it raises
    SequenceTooLong, which rl_step treats as a rejected sample."""
    from scanpaths_amd import hip
    from scanpaths_amd.utils.evaltools.scanmatch import MAX_SYMBOLS, SequenceTooLong
    sm = _sm(Xres=320, Yres=240, Xbin=16, Ybin=12, Offset=(0, 0), TempBin=50, Threshold=3.5)
    S = SO.submatrix(16, 12, 3.5)
    g = np.random.Generator(np.random.PCG64(5))
    limit = hip.lib().sp_scanmatch_max_len()
    long_fix = np.stack([g.uniform(0, 320, 12), g.uniform(0, 240, 12), g.uniform(100, 600, 12)], 1)
    long_fix[4, 2] = 230_000.0                               # one 230 s fixation -> 4600 symbols on its own
    short_fix = np.stack([g.uniform(0, 320, 7), g.uniform(0, 240, 7), g.uniform(100, 600, 7)], 1)
    seq, ln = sm.sequences([long_fix, short_fix])
    assert int(ln[0]) > limit and seq.shape[1] > limit
    got = sm.match_pairs(seq, ln, seq, ln, torch.tensor([(0, 1), (1, 0), (0, 0), (1, 1)], dtype=torch.int32)).cpu().numpy()
    a = SO.fixation_to_sequence(long_fix, 320, 240, 16, 12, (0, 0), 50.0).astype(np.int32)
    b = SO.fixation_to_sequence(short_fix, 320, 240, 16, 12, (0, 0), 50.0).astype(np.int32)
    want = [SO.nw_score(a, b, S, 0.0), SO.nw_score(b, a, S, 0.0), SO.nw_score(a, a, S, 0.0), SO.nw_score(b, b, S, 0.0)]
    assert list(got) == want
    huge = long_fix.copy()
    huge[4, 2] = 50.0 * (MAX_SYMBOLS + 10)
    with pytest.raises(SequenceTooLong):
        sm.sequences([huge])


def test_custom_mask_and_errors():
    sm = _sm(Xres=64, Yres=48, Xbin=4, Ybin=3, Threshold=1.5)
    g = np.random.Generator(np.random.PCG64(3))
    mask = g.integers(0, 12, (48, 64))
    sm.maskFromArray(mask)
    fix = np.stack([g.uniform(-3, 70, 9), g.uniform(-3, 52, 9)], 1)
    same(sm.fixationToSequence(fix), SO.fixation_to_sequence(fix, 64, 48, 4, 3, mask=mask), "custom mask")
    with pytest.raises(ValueError):
        sm.maskFromArray(np.zeros((3, 3)))
    with pytest.raises(IndexError):
        sm.match([0, 99], [1])
    with pytest.raises(IndexError):
        _sm(TempBin=50).sequences([np.zeros((2, 2))])


# ---- SED / STDE ------------------------------------------------------------------------------------------------
GOLD2 = np.load(os.path.join(os.path.dirname(__file__), "golden", "sed_stde.npz"))


def test_sed_stde_reference_self_check_and_random_pairs():
    """the reference's __main__ block (visual_attention_metrics.py:495-519) and all 40x40 random pairs (1..30 fixations):
    SED bit-exact; STDE within 4 ulp (numpy evaluation order is reproduced, exp() may differ in the last bit)"""
    from scanpaths_amd.utils.evaltools.visual_attention_metrics import (scaled_time_delay_embedding_similarity, sed_stde_pairs,
                                                                        string_edit_distance)
    ex = fixes("ex_fix")
    stim = np.zeros((768, 1024, 3), dtype=np.float32)
    assert string_edit_distance(stim, ex[0], ex[1]) == 9 == int(GOLD2["ex_sed"][0, 1])
    v = scaled_time_delay_embedding_similarity(ex[0], ex[1], stim)
    assert abs(v - 0.9064806433533912) <= 4 * np.spacing(0.9064806433533912)
    assert scaled_time_delay_embedding_similarity(ex[0][:0], ex[1], stim) is None
    pairs = [(i, j) for i in range(3) for j in range(3)]
    sed, stde = sed_stde_pairs(ex, pairs, stim.shape)
    assert np.array_equal(sed.cpu().numpy().reshape(3, 3), GOLD2["ex_sed"])
    want = GOLD2["ex_stde"]
    assert np.all(np.abs(stde.cpu().numpy().reshape(3, 3) - want) <= 4 * np.spacing(want))

    cat, off = GOLD2["rnd_fix"], GOLD2["rnd_fix_off"]
    fx = [cat[off[i]:off[i + 1]] for i in range(len(off) - 1)]
    n = len(fx)
    pairs = [(i, j) for i in range(n) for j in range(n)]
    sed, stde = sed_stde_pairs(fx, pairs, (240, 320, 3))
    assert np.array_equal(sed.cpu().numpy().reshape(n, n), GOLD2["rnd_sed"])
    got, want = stde.cpu().numpy().reshape(n, n), GOLD2["rnd_stde"]
    ulp = np.abs(got - want) / np.spacing(want)
    assert ulp.max() <= 4, ulp.max()
    sed8, _ = sed_stde_pairs(fx, [(i, j) for i in range(12) for j in range(12)], (240, 320, 3), n=8, want_stde=False)
    assert np.array_equal(sed8.cpu().numpy().reshape(12, 12), GOLD2["rnd_sed_n8"])
    with pytest.raises(ValueError):
        sed_stde_pairs([np.zeros((65, 2))], [(0, 0)], (240, 320, 3))
