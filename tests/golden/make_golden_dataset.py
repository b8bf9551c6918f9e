"""tests/golden/dataset_variants.npz + collate_f64.npz: outputs of the REAL reference datasets (build container only) on synthetic
files in a temporary directory, for SURVEY.md §8 row f4's remainder:

  * AiR.__getitem__ + collate_func under numpy-1.x DIVISION SEMANTICS (collate_f64.npz): the reference is pinned to numpy==1.19.2,
    where ``np.float32 scalar / python float`` is evaluated in float64; numpy 2.2 (installed here) keeps float32.  The downscale
    factors are ``origin_size / self.action_map[k]`` -- action_map is handed in as ints whose reflected division returns a
    np.float64 (a strong type under NEP 50), which makes every ``pos / downscale`` a float64 division exactly as under 1.19.2.
    (The ``duration_raw / 1000.0`` literal cannot be shimmed; durations are stored with the float64 result computed here from the
    reference's own float32 duration_raw -- values where the two roundings differ do not occur in these records, asserted.)
  * the same with blur_sigma = 1 (scipy gaussian_filter + renormalisation, AiR/dataset/dataset.py:144-146);
  * AiR_evaluation.__getitem__ + collate_func (:258-343): per-question grouping, fix_vectors, performances;
  * OSIE.__getitem__ (OSIE/dataset/dataset.py:59-115) and COCO_Search18.__getitem__ (COCO_Search18/dataset/dataset.py:88-175)
    incl. the detector-box attention map (images are made at the action-map size so that the stubbed skimage ``resize`` -- absent
    here -- is the identity and the map is the raw box raster divided by max + 1e-7).
Images are dummies (the transform returns zeros)."""
import importlib
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


class I(int):
    """an int whose use as a divisor yields np.float64 (numpy-1.x value-based result of float32_scalar / (int / int))"""
    def __rtruediv__(self, o):
        return np.float64(o) / np.float64(int(self))


def _shims():
    sk = types.ModuleType("skimage")
    sk.io = types.ModuleType("skimage.io")
    tr = types.ModuleType("skimage.transform")

    def resize(img, shape, **kw):
        assert tuple(img.shape) == tuple(int(s) for s in shape), "stub resize: identity only"
        return np.array(img, dtype=np.float64)
    tr.resize, tr.rescale, tr.downscale_local_mean = resize, None, None
    sk.transform = tr
    tv = types.ModuleType("torchvision")
    tv.transforms = types.ModuleType("torchvision.transforms")
    for name, mod in (("skimage", sk), ("skimage.io", sk.io), ("skimage.transform", tr), ("torchvision", tv),
                      ("torchvision.transforms", tv.transforms)):
        sys.modules.setdefault(name, mod)
    for name in ("tqdm", "matplotlib", "matplotlib.pyplot", "seaborn", "cv2"):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
    if not hasattr(sys.modules["tqdm"], "tqdm"):
        sys.modules["tqdm"].tqdm = lambda x, **k: x
    if "matplotlib.pyplot" in sys.modules and not hasattr(sys.modules["matplotlib"], "pyplot"):
        sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]


def load_dataset_module(task):
    for k in [k for k in sys.modules if k == "dataset" or k.startswith("dataset.")]:
        del sys.modules[k]
    sys.path.insert(0, f"/root/reference/{task}")
    try:
        return importlib.import_module("dataset.dataset")
    finally:
        sys.path.pop(0)


def air_records():
    from make_golden_collate import records
    recs = records()
    # several subjects per question for the evaluation grouping: questions q0, q1, q2 interleaved
    rng = np.random.Generator(np.random.PCG64(5))
    ev = []
    for i in range(8):
        n = int(rng.integers(1, 9))
        q = i % 3
        h, w = [(240, 320), (480, 640), (375, 500)][q]
        ts = np.cumsum(rng.uniform(50, 400, n))
        ans = ["yes", "no", "faild"][int(rng.integers(0, 3))]
        ev.append({"image_id": f"img{q}.jpg", "question_id": f"q{q:04d}", "height": h, "width": w, "length": n,
                   "X": [float(v) for v in rng.uniform(0, w - 1e-3, n)], "Y": [float(v) for v in rng.uniform(0, h - 1e-3, n)],
                   "T_start": [float(v) for v in ts], "T_end": [float(v) for v in ts + rng.uniform(80, 600, n)],
                   "subject_answer": ans, "answer": "yes"})
    return recs, ev


def flat(fvs):
    return (np.concatenate([np.stack([f["start_x"], f["start_y"], f["duration"]], 1) for f in fvs], 0), np.array([len(f) for f in fvs]))


def main():
    _shims()
    sys.path.insert(0, HERE)
    from PIL import Image
    zeros = lambda im: torch.zeros(3, 8, 8)
    out, out64 = {}, {}
    am = (I(30), I(40))
    # ---------------------------------------------------------------- AiR ------------------------------------------------------------
    D = load_dataset_module("AiR")
    recs, ev = air_records()
    with tempfile.TemporaryDirectory() as tmp:
        for d in ("img", "fix", "att"):
            os.makedirs(os.path.join(tmp, d))
        for r in recs + ev:
            Image.fromarray(np.zeros((8, 8, 3), dtype=np.uint8)).save(os.path.join(tmp, "img", r["image_id"]))
            np.save(os.path.join(tmp, "att", r["question_id"] + ".npy"), np.full((30, 40), 0.5))
        json.dump(recs, open(os.path.join(tmp, "fix", "AiR_fixations_train.json"), "w"))
        json.dump(ev, open(os.path.join(tmp, "fix", "AiR_fixations_validation.json"), "w"))
        for tag, sigma in (("", None), ("blur_", 1)):
            ds = D.AiR(os.path.join(tmp, "img"), os.path.join(tmp, "fix"), os.path.join(tmp, "att"), action_map=am, max_length=16,
                       blur_sigma=sigma, type="train", transform=zeros)
            batch = ds.collate_func([ds[i] for i in range(len(ds))])
            dst = out64 if tag == "" else out
            dst[tag + "scanpaths"] = batch["scanpaths"].numpy()
            dst[tag + "durations"] = batch["durations"].numpy()
            dst[tag + "action_masks"], dst[tag + "duration_masks"] = batch["action_masks"].numpy(), batch["duration_masks"].numpy()
        # numpy-1.x duration: float32(float64(duration_raw) / 1000.0); identical to the float32 division on these records
        for r, row in zip(recs, out64["durations"]):
            raw = np.array(r["T_end"]).astype(np.float32) - np.array(r["T_start"]).astype(np.float32)
            d64 = (raw.astype(np.float64) / 1000.0).astype(np.float32)[:16]
            assert np.array_equal(d64, row[:len(d64)]), "float64 and float32 duration roundings differ on a golden record"
        out64["records"] = np.frombuffer(json.dumps(recs).encode(), dtype=np.uint8)
        out["air_records"] = out64["records"]
        dse = D.AiR_evaluation(os.path.join(tmp, "img"), os.path.join(tmp, "fix"), os.path.join(tmp, "att"), action_map=(30, 40),
                               resize=(240, 320), type="validation", transform=zeros)
        eb = dse.collate_func([dse[i] for i in range(len(dse))])
        out["eval_records"] = np.frombuffer(json.dumps(ev).encode(), dtype=np.uint8)
        out["eval_fix"], out["eval_len"] = flat([f for l in eb["fix_vectors"] for f in l])
        out["eval_count"] = np.array([len(l) for l in eb["fix_vectors"]])
        out["eval_perf"] = np.array([int(p) for l in eb["performances"] for p in l])
        out["eval_qids"] = np.frombuffer(json.dumps(eb["question_ids"]).encode(), dtype=np.uint8)
        out["eval_imgs"] = np.frombuffer(json.dumps(eb["img_names"]).encode(), dtype=np.uint8)
    # ---------------------------------------------------------------- OSIE -----------------------------------------------------------
    D = load_dataset_module("OSIE")
    rng = np.random.Generator(np.random.PCG64(21))
    orecs = []
    for i, n in enumerate([1, 4, 16, 19, 7]):
        X, Y = rng.uniform(0, 800 - 1e-3, n), rng.uniform(0, 600 - 1e-3, n)
        if n >= 3:
            X[0], Y[0] = 0.0, 0.0
            X[1], Y[1] = 800 - 1e-3, 600 - 1e-3
            X[2], Y[2] = 20.0 * 7, 20.0 * 11                              # exact cell boundaries (800/40 = 600/30 = 20)
        orecs.append({"name": f"o{i}.jpg", "X": [float(v) for v in X], "Y": [float(v) for v in Y],
                      "T": [float(v) for v in rng.uniform(80, 700, n)], "length": n})
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "img"))
        os.makedirs(os.path.join(tmp, "fix"))
        for r in orecs:
            Image.fromarray(np.zeros((8, 8, 3), dtype=np.uint8)).save(os.path.join(tmp, "img", r["name"]))
        json.dump(orecs, open(os.path.join(tmp, "fix", "osie_fixations_train.json"), "w"))
        for tag, sigma in (("osie_", None), ("osie_blur_", 2)):
            ds = D.OSIE(os.path.join(tmp, "img"), os.path.join(tmp, "fix"), action_map=am, origin_size=(600, 800), max_length=16,
                        blur_sigma=sigma, type="train", transform=zeros)
            batch = ds.collate_func([ds[i] for i in range(len(ds))])
            for k in ("scanpaths", "durations", "action_masks", "duration_masks"):
                out[tag + k] = batch[k].numpy()
    out["osie_records"] = np.frombuffer(json.dumps(orecs).encode(), dtype=np.uint8)
    # ---------------------------------------------------------------- COCO-Search18 --------------------------------------------------
    D = load_dataset_module("COCO_Search18")
    rng = np.random.Generator(np.random.PCG64(33))
    crecs, dets = [], []
    tasks = ["bottle", "potted plant", "tv", "stop sign"]
    for i, n in enumerate([2, 6, 17, 9]):
        X, Y = rng.uniform(0, 512 - 1e-3, n), rng.uniform(0, 320 - 1e-3, n)
        X[0], Y[0] = 530.0, 340.0                                        # beyond the frame: clamped to 511 / 319 (:96-99)
        if n >= 3:
            X[1], Y[1] = 512.0, 320.0
            X[2], Y[2] = 12.8 * 5, 320 / 30 * 7
        crecs.append({"name": f"{1000 + i}.jpg", "task": tasks[i], "X": [float(v) for v in X], "Y": [float(v) for v in Y],
                      "T": [float(v) for v in rng.uniform(60, 500, n)], "length": n})
        for k in range(3):
            x0, y0 = int(rng.integers(0, 30)), int(rng.integers(0, 22))
            dets.append({"image_id": str(1000 + i), "category": tasks[(i + (k == 2)) % 4], "score": [0.9, 0.61, 0.95][k] if i != 3 else 0.3,
                         "bbox": [x0 + 0.7, y0 + 0.2, x0 + int(rng.integers(2, 9)) + 0.9, y0 + int(rng.integers(2, 7)) + 0.5]})
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "fix"))
        os.makedirs(os.path.join(tmp, "det"))
        for r in crecs:
            os.makedirs(os.path.join(tmp, "img", r["task"]), exist_ok=True)
            Image.fromarray(np.zeros((30, 40, 3), dtype=np.uint8)).save(os.path.join(tmp, "img", r["task"], r["name"]))
        json.dump(crecs, open(os.path.join(tmp, "fix", "coco_search18_fixations_TP_train_split1.json"), "w"))
        json.dump(dets, open(os.path.join(tmp, "det", "coco_search18_detector.json"), "w"))
        ds = D.COCO_Search18(os.path.join(tmp, "img"), os.path.join(tmp, "fix"), os.path.join(tmp, "det"), action_map=am, max_length=16,
                             blur_sigma=None, type="train", split="split1", transform=zeros, detector_threshold=0.6)
        batch = ds.collate_func([ds[i] for i in range(len(ds))])
        for k in ("scanpaths", "durations", "action_masks", "duration_masks", "attention_maps", "tasks"):
            out["coco_" + k] = batch[k].numpy()
    out["coco_records"] = np.frombuffer(json.dumps(crecs).encode(), dtype=np.uint8)
    out["coco_detector"] = np.frombuffer(json.dumps(dets).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "dataset_variants.npz"), **out)
    np.savez_compressed(os.path.join(HERE, "collate_f64.npz"), **out64)
    print({k: v.shape for k, v in out.items()})
    old = np.load(os.path.join(HERE, "collate.npz"))
    print("cells that differ between numpy-2 (float32) and numpy-1.x (float64) division:",
          int((old["scanpaths"] != out64["scanpaths"]).any(-1).sum()), "of", old["scanpaths"].shape[0] * 16, "steps")


if __name__ == "__main__":
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        main()
