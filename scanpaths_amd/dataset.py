"""Dataset-side targets on the device: what ``AiR.__getitem__`` + ``collate_func`` build per batch
(AiR/dataset/dataset.py:100-211) from the fixation records -- soft one-hot action targets ``scanpaths`` [B,T,1+Hm*Wm], ``durations``,
``action_masks``, ``duration_masks`` -- by ONE kernel launch over the ragged fixation lists (csrc/sampling.hip ``collate_kernel``)
instead of B python loops over T.  Image decoding / transforms and the attention-box resize (skimage) stay on the host: they need
the data files and libraries that are absent here; the attention map's ``/= max`` normalisation is offered (``normalise_attention``).

``blur_sigma`` (scipy gaussian_filter of every target map, default None in AiR/opts.py:14) is not implemented on the device:
pass blur_sigma=None or build those targets on the host."""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch

from . import hip
from .hip import check, ptr


def collate_targets(fixations: Sequence[dict], max_length: int = 16, action_map=(30, 40), device=None, f64_div: bool = False
                    ) -> Dict[str, torch.Tensor]:
    """fixations: records with "X", "Y", "T_start", "T_end" (sequences; ms) and "height", "width" (origin image size), as in
    the reference's fixation json (AiR/dataset/dataset.py:100-122).  f64_div: evaluate pixel -> cell in float64 (what numpy 1.x,
    the reference's pinned environment, does) instead of float32 (numpy >= 2, the environment the goldens were made in)."""
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
    return {"scanpaths": target, "durations": dur, "action_masks": am, "duration_masks": dm}


def normalise_attention(attention_maps: torch.Tensor) -> torch.Tensor:
    """attention_map /= attention_map.max() per sample (AiR/dataset/dataset.py:153), [B,1,Hm,Wm]"""
    return attention_maps / attention_maps.flatten(1).max(1).values.view(-1, 1, 1, 1)


def collate_func(samples: List[dict], max_length: int = 16, action_map=(30, 40), device=None) -> Dict[str, object]:
    """Batch assembly with the reference's keys (collate_func, :168-211).  Each sample: {"image" [3,H,W] tensor, "fixation":
    the fixation record, "attention_map" [1,Hm,Wm] (already resized), "img_name", "question_id"}; ``performance`` is derived as
    the reference does (:149): subject_answer == answer and subject_answer != "faild"."""
    t = collate_targets([s["fixation"] for s in samples], max_length, action_map, device)
    dev = t["scanpaths"].device
    data = dict(t)
    data["images"] = torch.stack([s["image"] for s in samples]).to(dev)
    data["attention_maps"] = torch.stack([torch.as_tensor(s["attention_map"], dtype=torch.float32) for s in samples]).to(dev)
    data["img_names"] = [s["img_name"] for s in samples]
    data["question_ids"] = [s["question_id"] for s in samples]
    data["performances"] = torch.tensor([(s["fixation"]["subject_answer"] == s["fixation"]["answer"]
                                          and s["fixation"]["subject_answer"] != "faild") for s in samples], device=dev)
    return data
