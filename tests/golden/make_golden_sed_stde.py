"""Golden vectors for SED / STDE from the REAL reference
(/root/reference/AiR/utils/evaltools/visual_attention_metrics.py:205-441), build container only:

    python tests/golden/make_golden_sed_stde.py

matplotlib / cv2 are imported by that module but not used by these two metrics; they are stubbed.
Writes tests/golden/sed_stde.npz: the reference's own self-check (the .mat example, stimulus 768x1024, :495-519) and seeded
random scanpath pairs in the evaluation configuration (stimulus 240x320x3, utils/evaluation.py:68-72)."""
import os
import sys
import types

import numpy as np
import scipy.io as sio

for name in ("matplotlib", "matplotlib.pyplot", "cv2"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
sys.path.insert(0, "/root/reference/AiR")
from utils.evaltools.visual_attention_metrics import (  # noqa: E402
    scaled_time_delay_embedding_similarity, string_edit_distance)

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    out = {}
    mat = sio.loadmat("/root/reference/OSIE/utils/evaltools/ScanMatch_DataExample.mat")
    ex = [np.asarray(mat[k], dtype=np.float64) for k in ("data1", "data2", "data3")]
    stim = np.zeros((768, 1024, 3), dtype=np.float32)
    out["ex_sed"] = np.array([[string_edit_distance(stim, a, b) for b in ex] for a in ex], dtype=np.int64)
    out["ex_stde"] = np.array([[scaled_time_delay_embedding_similarity(a, b, stim) for b in ex] for a in ex], dtype=np.float64)
    g = np.random.Generator(np.random.PCG64(424242))
    fixs = []
    for i in range(40):
        L = int(g.integers(1, 31)) if i % 5 else int(g.integers(1, 4))
        fixs.append(np.stack([g.uniform(0.0, 320.0, L), g.uniform(0.0, 240.0, L), g.uniform(50.0, 900.0, L)], 1))
    cat = np.concatenate(fixs, 0)
    off = np.zeros(len(fixs) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(f) for f in fixs])
    out["rnd_fix"], out["rnd_fix_off"] = cat, off
    stim = np.zeros((240, 320, 3), dtype=np.float32)
    n = len(fixs)
    out["rnd_sed"] = np.array([[string_edit_distance(stim, fixs[i], fixs[j]) for j in range(n)] for i in range(n)], dtype=np.int64)
    out["rnd_stde"] = np.array([[scaled_time_delay_embedding_similarity(fixs[i], fixs[j], stim) for j in range(n)]
                                for i in range(n)], dtype=np.float64)
    out["rnd_sed_n8"] = np.array([[string_edit_distance(stim, fixs[i], fixs[j], n=8) for j in range(12)] for i in range(12)],
                                 dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "sed_stde.npz"), **out)
    print("example SED", out["ex_sed"].tolist())
    print("example STDE", out["ex_stde"].tolist())
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
