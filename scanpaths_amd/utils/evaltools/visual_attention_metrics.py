"""SED and STDE with the reference's interface (utils/evaltools/visual_attention_metrics.py:300-318, :392-441), computed by
csrc/scanmetrics.hip, plus the batched form the validation loops need:

    sed  = string_edit_distance(stimulus, human_scanpath, simulated_scanpath)                 # int
    stde = scaled_time_delay_embedding_similarity(human_scanpath, simulated_scanpath, stimulus)   # float, None if a path is empty
    sed, stde = sed_stde_pairs(scanpaths, pairs, stimulus.shape)                                  # device tensors [npairs]

SED is bit-exact; STDE follows numpy's float64 evaluation order (differences only in the last bit of exp()).  No CPU path."""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from ... import hip
from ...hip import check, ptr


def _device() -> torch.device:
    if not torch.cuda.is_available():
        raise hip.HipError("scanpaths_amd scanpath metrics run on a HIP device only (no CPU path)")
    return torch.device("cuda", torch.cuda.current_device())


def sed_stde_pairs(scanpaths: Sequence[np.ndarray], pairs, image_shape, n: int = 5, want_sed: bool = True, want_stde: bool = True
                   ) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    """scanpaths: list of [n_k, >=2] arrays (x, y, ...); pairs: int [npairs, 2] = (human index, simulated index);
    image_shape: shape of the stimulus (height, width[, channels]).  Returns (sed int32 [npairs], stde float64 [npairs])."""
    dev = _device()
    L = hip.lib()
    arrs = [np.asarray(a, dtype=np.float64).reshape(len(a), -1) for a in scanpaths]
    ncol = arrs[0].shape[1]
    if any(a.shape[1] != ncol for a in arrs) or ncol < 2:
        raise ValueError("scanpaths need the same number (>= 2) of columns")
    counts = [a.shape[0] for a in arrs]
    if max(counts) > L.sp_scan_max_fixations():
        raise ValueError(f"scanpath of {max(counts)} fixations exceeds the kernel limit {L.sp_scan_max_fixations()}")
    count = torch.tensor(counts, dtype=torch.int32)
    start = (torch.cumsum(count.to(torch.int64), 0) - count.to(torch.int64)).to(dev)
    cat = np.concatenate(arrs, 0)
    if cat.shape[0] == 0:
        cat = np.zeros((1, ncol))
    fix = torch.from_numpy(cat).to(dev)
    pr = torch.as_tensor(pairs, dtype=torch.int32).reshape(-1, 2).to(dev).contiguous()
    npairs = pr.shape[0]
    sed = torch.empty(npairs, dtype=torch.int32, device=dev) if want_sed else None
    stde = torch.empty(npairs, dtype=torch.float64, device=dev) if want_stde else None
    count_d = count.to(dev)          # named: must outlive the launch
    if npairs:
        check(L.sp_scan_sed_stde(ptr(fix), ncol, ptr(start), ptr(count_d), ptr(pr), npairs, int(image_shape[0]),
                                 int(image_shape[1]), int(n), float(max(image_shape)), ptr(sed), ptr(stde), hip.stream()),
              "sp_scan_sed_stde")
    return sed, stde


def string_edit_distance(stimulus, human_scanpath, simulated_scanpath, n=5, substitution_cost=1, msg=False):
    # substitution_cost is accepted and ignored, as in the reference (:317 calls _Levenshtein without it)
    sed, _ = sed_stde_pairs([human_scanpath, simulated_scanpath], [(0, 1)], np.shape(stimulus), n=n, want_stde=False)
    return int(sed.item())


def scaled_time_delay_embedding_similarity(human_scanpath, simulated_scanpath, image, toPlot=False, msg=False):
    if len(human_scanpath) == 0 or len(simulated_scanpath) == 0:
        return None
    _, stde = sed_stde_pairs([human_scanpath, simulated_scanpath], [(0, 1)], np.shape(image), want_sed=False)
    return float(stde.item())
