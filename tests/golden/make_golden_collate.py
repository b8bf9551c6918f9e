"""tests/golden/collate.npz: outputs of the REAL reference's ``AiR.__getitem__`` + ``collate_func`` (AiR/dataset/dataset.py:100-211)
on synthetic data files written to a temporary directory (build container only).  Stored: the fixation records (inputs) and the
reference's target_scanpath / duration / action_mask / duration_mask / performances (outputs); images and attention boxes are
dummies (the image transform returns zeros; attention boxes are generated at the action-map size so that skimage's ``resize``
-- absent here, stubbed by a same-shape identity -- does not take part).  Run with numpy >= 2 (float32 / weak-python-float
division semantics, see oracle/sampling_oracle.py)."""
import importlib
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/AiR"


def _shims():
    sk = types.ModuleType("skimage")
    sk.io = types.ModuleType("skimage.io")
    tr = types.ModuleType("skimage.transform")

    def resize(img, shape):
        assert tuple(img.shape) == tuple(shape), "stub resize: identity only"
        return np.array(img, dtype=np.float64)
    tr.resize, tr.rescale, tr.downscale_local_mean = resize, None, None
    sk.transform = tr
    tv = types.ModuleType("torchvision")
    tv.transforms = types.ModuleType("torchvision.transforms")
    for name, mod in (("skimage", sk), ("skimage.io", sk.io), ("skimage.transform", tr), ("torchvision", tv),
                      ("torchvision.transforms", tv.transforms)):
        sys.modules.setdefault(name, mod)


def records():
    rng = np.random.Generator(np.random.PCG64(11))
    recs = []
    sizes = [(240, 320), (480, 640), (320, 512), (375, 500), (600, 800)]
    for i, n in enumerate([0, 1, 5, 15, 16, 17, 23, 9, 3, 12]):
        h, w = sizes[i % len(sizes)]
        X = rng.uniform(0, w - 1e-3, n)
        Y = rng.uniform(0, h - 1e-3, n)
        if n >= 3:      # exact cell boundaries and the last pixel
            X[0], Y[0] = 0.0, 0.0
            X[1], Y[1] = w - 1e-3, h - 1e-3
            X[2], Y[2] = (w / 40) * 5, (h / 30) * 7
        ts = np.cumsum(rng.uniform(50, 400, n)) if n else np.zeros(0)
        te = ts + rng.uniform(80, 600, n)
        ans = ["yes", "no", "faild"][i % 3]
        recs.append({"image_id": f"img{i}.jpg", "question_id": f"q{i:04d}", "height": h, "width": w,
                     "X": [float(v) for v in X], "Y": [float(v) for v in Y],
                     "T_start": [float(v) for v in ts], "T_end": [float(v) for v in te],
                     "subject_answer": ans, "answer": "yes" if i % 2 == 0 else ans})
    return recs


if __name__ == "__main__":
    _shims()
    sys.path.insert(0, REF)
    D = importlib.import_module("dataset.dataset")
    from PIL import Image
    recs = records()
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "img"))
        os.makedirs(os.path.join(tmp, "fix"))
        os.makedirs(os.path.join(tmp, "att"))
        for r in recs:
            Image.fromarray(np.zeros((8, 8, 3), dtype=np.uint8)).save(os.path.join(tmp, "img", r["image_id"]))
            rng = np.random.Generator(np.random.PCG64(abs(hash(r["question_id"])) % 1000))
            np.save(os.path.join(tmp, "att", r["question_id"] + ".npy"), rng.random((30, 40)))
        with open(os.path.join(tmp, "fix", "AiR_fixations_train.json"), "w") as f:
            json.dump(recs, f)
        ds = D.AiR(os.path.join(tmp, "img"), os.path.join(tmp, "fix"), os.path.join(tmp, "att"), action_map=(30, 40),
                   max_length=16, blur_sigma=None, type="train", transform=lambda im: torch.zeros(3, 8, 8))
        batch = ds.collate_func([ds[i] for i in range(len(ds))])
    out = {"records": np.frombuffer(json.dumps(recs).encode(), dtype=np.uint8),
           "scanpaths": batch["scanpaths"].numpy(), "durations": batch["durations"].numpy(),
           "action_masks": batch["action_masks"].numpy(), "duration_masks": batch["duration_masks"].numpy(),
           "performances": batch["performances"].numpy(), "numpy_version": np.frombuffer(np.__version__.encode(), dtype=np.uint8)}
    np.savez_compressed(os.path.join(HERE, "collate.npz"), **out)
    print("wrote collate.npz", {k: v.shape for k, v in out.items()})
