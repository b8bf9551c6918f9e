"""Synthetic batches with the shapes/statistics of the reference datasets.

There is no dataset on the GPU box; these generators follow SURVEY.md §8(d):
  images          N(0,1) f32 [B,3,H,W]  (post-ImageNet-normalisation scale, AiR/train.py:43-47)
  attention_maps  U(0,1)/max per sample [B,1,Hm,Wm]   (AiR/dataset/dataset.py:151-154)
  performances    Bernoulli(0.5) bool [B]             (AiR/dataset/dataset.py, "performances")
  tasks           uniform int64 in [0,18)             (COCO_Search18)
  scanpaths       [B,T,A] one-hot: cell index+1 for t<L, index 0 (terminate) for t>=L
                  (AiR/dataset/dataset.py:139-147, blur_sigma=None)
  action_masks    1 for t<=L (t<T); duration_masks 1 for t<L   (dataset.py:133-136)
  durations       U(0.1,0.6) seconds
All draws come from numpy PCG64 so a (seed, rank) pair gives identical bytes everywhere.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch


LENGTH_LAWS = ("uniform_1_T", "uniform_halfT_T", "full_T")


def make_batch(task: str, batch: int, height: int, width: int, T: int, seed: int = 0,
               rank: int = 0, length_law: str = "uniform_1_T") -> Dict[str, torch.Tensor]:
    """length_law: distribution of the scanpath length L (the only thing the masked-step sparsity of the backward pass depends on).
    "uniform_1_T" = SURVEY.md 8(d)'s contract, L ~ U{1..T}; "uniform_halfT_T" (L ~ U{ceil(T/2)..T}) and "full_T" (L = T) are bench.py's
    sensitivity legs -- every other draw is identical for the same (seed, rank)."""
    assert task in ("AiR", "OSIE", "COCO_Search18")
    assert length_law in LENGTH_LAWS, length_law
    rng = np.random.Generator(np.random.PCG64(seed * 1000003 + rank))
    Hm, Wm = height // 8, width // 8
    P = Hm * Wm
    A = P + 1
    out: Dict[str, torch.Tensor] = {}
    out["images"] = torch.from_numpy(rng.standard_normal((batch, 3, height, width), dtype=np.float32))
    att = rng.random((batch, 1, Hm, Wm), dtype=np.float32)
    att /= att.reshape(batch, -1).max(axis=1).reshape(batch, 1, 1, 1)
    out["attention_maps"] = torch.from_numpy(att)
    out["performances"] = torch.from_numpy(rng.random(batch) < 0.5)
    out["tasks"] = torch.from_numpy(rng.integers(0, 18, size=batch, dtype=np.int64))
    L = rng.integers(1, T + 1, size=batch)
    if length_law == "uniform_halfT_T":      # (drawn from the same stream position: the other tensors do not change with the law)
        lo = (T + 1) // 2
        L = lo + (L - 1) * (T - lo + 1) // T
    elif length_law == "full_T":
        L = np.full(batch, T)
    cells = rng.integers(0, P, size=(batch, T))
    scan = np.zeros((batch, T, A), dtype=np.float32)
    amask = np.zeros((batch, T), dtype=np.float32)
    dmask = np.zeros((batch, T), dtype=np.float32)
    for b in range(batch):
        for t in range(T):
            if t < L[b]:
                scan[b, t, cells[b, t] + 1] = 1.0
                dmask[b, t] = 1.0
            else:
                scan[b, t, 0] = 1.0
            if t <= L[b]:
                amask[b, t] = 1.0
    out["scanpaths"] = torch.from_numpy(scan)
    out["action_masks"] = torch.from_numpy(amask)
    out["duration_masks"] = torch.from_numpy(dmask)
    out["durations"] = torch.from_numpy(rng.uniform(0.1, 0.6, size=(batch, T)).astype(np.float32))
    return out
