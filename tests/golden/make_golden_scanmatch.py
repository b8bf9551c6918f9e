"""Golden vectors for the ScanMatch scorer, produced by the REAL reference here
(/root/reference/AiR/utils/evaltools/scanmatch.py, GazeParser-derived) -- run in the build container only:

    python tests/golden/make_golden_scanmatch.py

Writes tests/golden/scanmatch.npz:
  * the three example scanpaths of the reference's own self-check fixture (OSIE/utils/evaltools/ScanMatch_DataExample.mat,
    converted to plain arrays -- data, not code) with the sequences / scores / F matrix / alignment the reference
    computes for them (its __main__ block, scanmatch.py:222-257);
  * seeded random scanpaths in the evaluation configuration (utils/evaluation.py:22-23: 320x240, 16x12 bins, TempBin 50,
    threshold 3.5) incl. out-of-range coordinates, zero-length results, and a non-zero GapValue: all-pairs scores.
Ragged data is stored as (concatenated values, offsets)."""
import os
import sys
import warnings

import numpy as np
import scipy.io as sio

sys.path.insert(0, "/root/reference/AiR")
from utils.evaltools.scanmatch import ScanMatch   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def ragged(lst, dtype):
    off = np.zeros(len(lst) + 1, dtype=np.int64)
    for i, a in enumerate(lst):
        off[i + 1] = off[i] + len(a)
    cat = np.concatenate([np.asarray(a, dtype=dtype).reshape(-1) for a in lst]) if lst else np.zeros(0, dtype)
    return cat.astype(dtype), off


def all_pairs(sm, seqs):
    n = len(seqs)
    out = np.zeros((n, n), dtype=np.float64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i in range(n):
            for j in range(n):
                out[i, j] = sm.match(seqs[i], seqs[j])[0]
    return out


def main():
    out = {}
    # ---- the reference's own example -------------------------------------------------------------------------
    mat = sio.loadmat("/root/reference/OSIE/utils/evaltools/ScanMatch_DataExample.mat")
    ex = [np.asarray(mat[k], dtype=np.float64) for k in ("data1", "data2", "data3")]
    cat, off = ragged([e.reshape(-1) for e in ex], np.float64)
    out["ex_fix"], out["ex_fix_off"] = cat, off // 3
    cfg = dict(Xres=1024, Yres=768, Xbin=12, Ybin=8, Offset=(0, 0), Threshold=3.5)
    wd, wod = ScanMatch(TempBin=100, **cfg), ScanMatch(**cfg)
    s_wd = [wd.fixationToSequence(e).astype(np.int32) for e in ex]
    s_wod = [wod.fixationToSequence(e[:, :2]).astype(np.int32) for e in ex]
    out["ex_seq_wd"], out["ex_seq_wd_off"] = ragged(s_wd, np.int32)
    out["ex_seq_wod"], out["ex_seq_wod_off"] = ragged(s_wod, np.int32)
    out["ex_scores_wd"] = all_pairs(wd, s_wd)
    out["ex_scores_wod"] = all_pairs(wod, s_wod)
    score, align, F = wd.match(s_wd[0], s_wd[1])
    out["ex_match01_score"], out["ex_match01_align"], out["ex_match01_F"] = np.float64(score), align, F
    score, align, F = wod.match(s_wod[1], s_wod[2])
    out["ex_match12_wod_score"], out["ex_match12_wod_align"], out["ex_match12_wod_F"] = np.float64(score), align, F
    out["ex_submatrix"] = wd.SubMatrix
    out["ex_mask_row0"], out["ex_mask_col0"] = wd.mask[0].copy(), wd.mask[:, 0].copy()

    # ---- random scanpaths, evaluation configuration -------------------------------------------------------------
    g = np.random.Generator(np.random.PCG64(20260102))
    fixs = []
    for i in range(28):
        L = int(g.integers(1, 17))
        x = g.uniform(-12.0, 335.0, L)
        y = g.uniform(-8.0, 248.0, L)
        d = g.uniform(0.0, 1400.0, L)
        if i == 5:
            d[:] = g.uniform(0.0, 24.0, L)        # every fixation rounds to zero repeats -> empty sequence with TempBin 50
        if i == 6:
            d[:] = 25.0                            # exactly half a bin: round-half-even
        if i == 7:
            d[:] = 75.0
        fixs.append(np.stack([x, y, d], 1))
    cat, off = ragged([f.reshape(-1) for f in fixs], np.float64)
    out["rnd_fix"], out["rnd_fix_off"] = cat, off // 3
    cfg = dict(Xres=320, Yres=240, Xbin=16, Ybin=12, Offset=(0, 0), Threshold=3.5)
    wd, wod = ScanMatch(TempBin=50, **cfg), ScanMatch(**cfg)
    s_wd = [wd.fixationToSequence(f).astype(np.int32) for f in fixs]
    s_wod = [wod.fixationToSequence(f).astype(np.int32) for f in fixs]
    out["rnd_seq_wd"], out["rnd_seq_wd_off"] = ragged(s_wd, np.int32)
    out["rnd_seq_wod"], out["rnd_seq_wod_off"] = ragged(s_wod, np.int32)
    out["rnd_scores_wd"] = all_pairs(wd, s_wd)
    out["rnd_scores_wod"] = all_pairs(wod, s_wod)
    # non-zero gap penalty and an offset
    cfg2 = dict(Xres=300, Yres=200, Xbin=10, Ybin=7, Offset=(10, 20), Threshold=2.0, GapValue=-0.75)
    g2 = ScanMatch(TempBin=80, **cfg2)
    s_g = [g2.fixationToSequence(f).astype(np.int32) for f in fixs]
    out["gap_seq"], out["gap_seq_off"] = ragged(s_g, np.int32)
    out["gap_scores"] = all_pairs(g2, s_g)
    score, align, F = g2.match(s_g[0], s_g[1])
    out["gap_match01_align"], out["gap_match01_F"] = align, F
    np.savez_compressed(os.path.join(HERE, "scanmatch.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})
    print("example scores wd", out["ex_scores_wd"][0, 1], out["ex_scores_wd"][0, 2], out["ex_scores_wd"][1, 2])
    print("example scores wod", out["ex_scores_wod"][0, 1], out["ex_scores_wod"][0, 2], out["ex_scores_wod"][2, 2])


if __name__ == "__main__":
    main()
