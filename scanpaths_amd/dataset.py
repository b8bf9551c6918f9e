"""Dataset-side targets on the device: what ``AiR.__getitem__`` + ``collate_func`` build per batch
(AiR/dataset/dataset.py:100-211) from the fixation records -- soft one-hot action targets ``scanpaths`` [B,T,1+Hm*Wm], ``durations``,
``action_masks``, ``duration_masks`` -- by ONE kernel launch over the ragged fixation lists (csrc/sampling.hip ``collate_kernel``)
instead of B python loops over T.  Image decoding / transforms and the attention-box resize (skimage) stay on the host: they need
the data files and libraries that are absent here; the attention map's ``/= max`` normalisation is offered (``normalise_attention``).

Round 3 adds the rest of the reference's dataset surface that does not need image files or skimage:
  * ``blur_sigma`` targets (gaussian_filter of every target map + renormalisation, :144-147) by a second launch (``sp_blur_targets``);
  * the OSIE (OSIE/dataset/dataset.py:59-115: fixed 600x800 origin, durations in "T") and COCO-Search18
    (COCO_Search18/dataset/dataset.py:88-128: fixed 320x512 origin, out-of-range fixations clamped to the last pixel) variants;
  * the evaluation / RL datasets' per-question grouping (``AiR_evaluation`` / ``AiR_rl.__getitem__`` + ``collate_func``,
    AiR/dataset/dataset.py:258-343, 390-473; the COCO and OSIE evaluation sets likewise): ragged host-side lists of the reference's
    structured fixation arrays -- they feed host-side consumers (utils/evaluation.py, the RL reward), so they stay on the host;
  * the COCO detector-box attention map before its skimage resize (COCO_Search18/dataset/dataset.py:150-160) and the
    normalisations behind the resize (AiR ``/= max``, COCO ``/= max + 1e-7``)."""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch

from . import hip
from .hip import check, ptr


def collate_targets(fixations: Sequence[dict], max_length: int = 16, action_map=(30, 40), device=None, f64_div: bool = True,
                    blur_sigma=None) -> Dict[str, torch.Tensor]:
    """fixations: records with "X", "Y", "T_start", "T_end" (sequences; ms) and "height", "width" (origin image size), as in
    the reference's fixation json (AiR/dataset/dataset.py:100-122).  f64_div (default): evaluate pixel -> cell and ms -> s in
    float64, what the reference's pinned numpy==1.19.2 does with ``float32_scalar / python_float`` (value-based casting;
    tests/golden/collate_f64.npz); False: float32, what numpy >= 2 does with the same source (tests/golden/collate.npz).
    blur_sigma: the reference's ``gaussian_filter`` + renormalisation of every target map (:144-147; None = one-hot targets)."""
    if device is None:
        if not torch.cuda.is_available():
            raise hip.HipError("scanpaths_amd.dataset.collate_targets runs on a HIP device only (no CPU path)")
        device = torch.device("cuda", torch.cuda.current_device())
    B = len(fixations)
    if B == 0:
        raise ValueError("empty batch")
    cnt = np.array([len(f["X"]) for f in fixations], dtype=np.int32)
    start = np.cumsum(cnt, dtype=np.int64) - cnt
    cat = lambda k: np.concatenate([np.asarray(f[k], dtype=np.float64).astype(np.float32).reshape(-1) for f in fixations] + [np.zeros(1, np.float32)])
    X, Y, Ts, Te = (torch.from_numpy(cat(k)).to(device) for k in ("X", "Y", "T_start", "T_end"))
    ow = torch.tensor([float(f["width"]) for f in fixations], dtype=torch.float64, device=device)
    oh = torch.tensor([float(f["height"]) for f in fixations], dtype=torch.float64, device=device)
    Hm, Wm = int(action_map[0]), int(action_map[1])
    T = int(max_length)
    target = torch.empty((B, T, 1 + Hm * Wm), dtype=torch.float32, device=device)
    dur = torch.empty((B, T), dtype=torch.float32, device=device)
    am = torch.empty((B, T), dtype=torch.float32, device=device)
    dm = torch.empty((B, T), dtype=torch.float32, device=device)
    start_d, cnt_d = torch.from_numpy(start).to(device), torch.from_numpy(cnt).to(device)     # named: must outlive the launch
    check(hip.lib().sp_collate_targets(ptr(X), ptr(Y), ptr(Ts), ptr(Te), ptr(start_d), ptr(cnt_d), ptr(ow), ptr(oh), B, T, Hm,
                                       Wm, int(f64_div), ptr(target), ptr(dur), ptr(am), ptr(dm), hip.stream()),
          "sp_collate_targets")
    if blur_sigma:
        check(hip.lib().sp_blur_targets(ptr(target), B * T, Hm, Wm, float(blur_sigma), hip.stream()), "sp_blur_targets")
    return {"scanpaths": target, "durations": dur, "action_masks": am, "duration_masks": dm}


def _with_durations(fixations: Sequence[dict], origin_size, clamp_to=None) -> List[dict]:
    """OSIE / COCO-Search18 records ("X", "Y", "T" = duration in ms, one fixed origin size) in the form collate_targets takes:
    T_start = 0, T_end = T (T - 0 is exact in float32).  clamp_to (COCO, :96-99): positions >= map * downscale -> that minus 1,
    evaluated in float32 like the reference's masked assignment on float32 arrays."""
    out = []
    for f in fixations:
        X = np.asarray(f["X"], dtype=np.float64).astype(np.float32)
        Y = np.asarray(f["Y"], dtype=np.float64).astype(np.float32)
        if clamp_to is not None:
            lim_x, lim_y = clamp_to
            X = X.copy()
            Y = Y.copy()
            X[X >= lim_x] = lim_x - 1
            Y[Y >= lim_y] = lim_y - 1
        T = np.asarray(f["T"], dtype=np.float64).astype(np.float32)
        out.append({"X": X, "Y": Y, "T_start": np.zeros_like(T), "T_end": T, "height": origin_size[0], "width": origin_size[1]})
    return out


def collate_targets_osie(fixations: Sequence[dict], max_length: int = 16, action_map=(30, 40), origin_size=(600, 800), device=None,
                         f64_div: bool = True, blur_sigma=None) -> Dict[str, torch.Tensor]:
    """OSIE.__getitem__ targets (OSIE/dataset/dataset.py:59-115): records {"X", "Y", "T"}, fixed origin_size"""
    return collate_targets(_with_durations(fixations, origin_size), max_length, action_map, device, f64_div, blur_sigma)


def collate_targets_coco(fixations: Sequence[dict], max_length: int = 16, action_map=(30, 40), device=None, f64_div: bool = True,
                         blur_sigma=None) -> Dict[str, torch.Tensor]:
    """COCO_Search18.extract_scanpath_info (COCO_Search18/dataset/dataset.py:88-128): records {"X", "Y", "T"} in the 320x512
    frame; positions at or beyond map * downscale are clamped to that minus one pixel (:96-99)."""
    Hm, Wm = action_map
    lim = (Wm * (512 / Wm), Hm * (320 / Hm))                       # self.action_map[1] * self.downscale_x, ... (python floats)
    return collate_targets(_with_durations(fixations, (320, 512), clamp_to=lim), max_length, action_map, device, f64_div, blur_sigma)


COCO_OBJECT_NAMES = ["bottle", "bowl", "car", "chair", "clock", "cup", "fork", "keyboard", "knife", "laptop", "microwave", "mouse",
                     "oven", "potted plant", "sink", "stop sign", "toilet", "tv"]


def index_detections(detector: Sequence[dict], threshold: float = 0.6) -> Dict[str, List[dict]]:
    """imgs_2_det of the reference's constructor (COCO_Search18/dataset/dataset.py:66-69): detections of the 18 search categories
    with score >= threshold, per image id"""
    out: Dict[str, List[dict]] = {}
    for det in detector:
        if det["category"] in COCO_OBJECT_NAMES and det["score"] >= threshold:
            out.setdefault(det["image_id"], []).append(det)
    return out


def detector_box_map(dets: Sequence[dict], task: str, det_size) -> np.ndarray:
    """The binary map of the detector boxes of the searched category at the detector image's size (:150-158); the reference then
    resizes it with skimage (absent here: host-side, out of scope) and normalises it (``normalise_attention(eps=1e-7)``)."""
    m = np.zeros((int(det_size[0]), int(det_size[1])), dtype=np.float32)
    for det in dets:
        if det["category"] == task:
            x_min, y_min, x_max, y_max = (int(det["bbox"][k]) for k in range(4))
            m[y_min:y_max, x_min:x_max] = 1
    return m


def fixation_vectors(fixation: dict, resize=(240, 320), origin_size=None) -> np.ndarray:
    """One record -> the reference's structured fixation array (start_x, start_y, duration: f8) in the ``resize`` frame
    (AiR/dataset/dataset.py:275-293; COCO_Search18/dataset/dataset.py:291-305 with the fixed 320x512 origin and "T").  The
    divisions are float32-array / python-float (float32 under every numpy), then stored as f8."""
    oh, ow = origin_size if origin_size is not None else (fixation["height"], fixation["width"])
    rx, ry = ow / resize[1], oh / resize[0]
    x = np.array(fixation["X"]).astype(np.float32) / rx
    y = np.array(fixation["Y"]).astype(np.float32) / ry
    if "T" in fixation:
        d = np.array(fixation["T"]).astype(np.float32) / 1000.0
    else:
        d = (np.array(fixation["T_end"]).astype(np.float32) - np.array(fixation["T_start"]).astype(np.float32)) / 1000.0
    n = fixation["length"]
    fv = np.zeros(n, dtype={"names": ("start_x", "start_y", "duration"), "formats": ("f8", "f8", "f8")})
    fv["start_x"], fv["start_y"], fv["duration"] = x[:n], y[:n], d[:n]
    return fv


def group_by_question(fixations: Sequence[dict], resize=(240, 320)) -> List[dict]:
    """AiR_evaluation / AiR_rl (AiR/dataset/dataset.py:236-241, 258-305): one sample per question id in first-seen order, holding
    every subject's fixation vector and performance flag (subject_answer == answer and != "faild")."""
    order, groups = [], {}
    for f in fixations:
        q = f["question_id"]
        if q not in groups:
            groups[q] = {"question_id": q, "img_name": f["image_id"], "fix_vectors": [], "performances": []}
            order.append(q)
        groups[q]["img_name"] = f["image_id"]                        # (:240) the last record of a question names the image
        groups[q]["fix_vectors"].append(fixation_vectors(f, resize))
        groups[q]["performances"].append(f["subject_answer"] == f["answer"] and f["subject_answer"] != "faild")
    return [groups[q] for q in order]


def collate_evaluation(samples: List[dict], device=None) -> Dict[str, object]:
    """collate_func of AiR_evaluation / AiR_rl (:307-343, :437-473): images and attention maps stacked, the ragged
    fix_vectors / performances kept as per-image lists.  Each sample: group_by_question's dict + "image" [3,H,W] and
    "attention_map" [1,Hm,Wm]."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    return {"images": torch.stack([s["image"] for s in samples]).to(device),
            "fix_vectors": [s["fix_vectors"] for s in samples],
            "attention_maps": torch.stack([torch.as_tensor(s["attention_map"], dtype=torch.float32) for s in samples]).to(device),
            "img_names": [s["img_name"] for s in samples],
            "performances": [s["performances"] for s in samples],
            "question_ids": [s["question_id"] for s in samples]}


def normalise_attention(attention_maps: torch.Tensor, eps: float = 0.0) -> torch.Tensor:
    """attention_map /= attention_map.max() per sample (AiR/dataset/dataset.py:153), [B,1,Hm,Wm]; COCO-Search18 divides by
    max + 1e-7 (COCO_Search18/dataset/dataset.py:159: eps=1e-7, an image without a detection stays all-zero)"""
    return attention_maps / (attention_maps.flatten(1).max(1).values.view(-1, 1, 1, 1) + eps)


def collate_func(samples: List[dict], max_length: int = 16, action_map=(30, 40), device=None, f64_div: bool = True,
                 blur_sigma=None) -> Dict[str, object]:
    """Batch assembly with the reference's keys (collate_func, :168-211).  Each sample: {"image" [3,H,W] tensor, "fixation":
    the fixation record, "attention_map" [1,Hm,Wm] (already resized), "img_name", "question_id"}; ``performance`` is derived as
    the reference does (:149): subject_answer == answer and subject_answer != "faild"."""
    t = collate_targets([s["fixation"] for s in samples], max_length, action_map, device, f64_div, blur_sigma)
    dev = t["scanpaths"].device
    data = dict(t)
    data["images"] = torch.stack([s["image"] for s in samples]).to(dev)
    data["attention_maps"] = torch.stack([torch.as_tensor(s["attention_map"], dtype=torch.float32) for s in samples]).to(dev)
    data["img_names"] = [s["img_name"] for s in samples]
    data["question_ids"] = [s["question_id"] for s in samples]
    data["performances"] = torch.tensor([(s["fixation"]["subject_answer"] == s["fixation"]["answer"]
                                          and s["fixation"]["subject_answer"] != "faild") for s in samples], device=dev)
    return data
