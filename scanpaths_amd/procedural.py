"""Platform-independent procedural parameter generator.

No network exists here or on the GPU box, so neither the ImageNet ResNet weights
(reference: ``models/resnet.py:13,186-187`` downloads them) nor the authors'
checkpoints are available.  Parity tests therefore fill the reference (in the
survey container), the oracle and the HIP model with *the same* numbers produced
by this generator: every tensor is drawn from ``numpy.random.Generator(PCG64)``
seeded by ``crc32(key) ^ seed``, so the value of a tensor depends only on its
state_dict key, its shape and the seed -- not on key order, torch version or
device.

The distributions are chosen so a random-init network is numerically
non-degenerate through 16 recurrent steps (signals neither vanish nor explode):
He-normal for ``resnet.*`` convolutions (reference ``models/resnet.py:112-118``),
fan-in scaled normal elsewhere, non-trivial BatchNorm affine/running statistics
so eval-mode BN is actually exercised.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Mapping, Tuple

import numpy as np
import torch


def _rng(key: str, seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF))


FAMILIES = ("default", "tame", "tame_sharp")
TAME_LAST_BN_GAIN = 0.25
SHARP_HEAD_GAIN = 8.0          # family "tame_sharp": gain of the two layers that emit the action logits


def _is_last_bn_of_block(prefix: str, bn_keys) -> bool:
    """resnet.<4..7>.<i>.bn3 (Bottleneck) or .bn2 of a BasicBlock (no bn3 sibling): the BatchNorm that closes the residual branch"""
    head, _, leaf = prefix.rpartition(".")
    if not prefix.startswith("resnet.") or leaf not in ("bn2", "bn3"):
        return False
    return leaf == "bn3" or (head + ".bn3") not in bn_keys


def procedural_tensor(key: str, shape: Tuple[int, ...], bn_keys: Iterable[str], seed: int = 0,
                      family: str = "default") -> torch.Tensor:
    """Value for one state_dict entry. ``bn_keys`` = set of module prefixes that are BatchNorm.

    family "default": the round-1 distributions (below).  In eval mode its random BatchNorm statistics do not normalise, every
    residual block doubles the variance, the encoder output reaches rms ~1e4 and the decoder gates become step functions: the
    16-step recurrence is chaotic (the reference's own fp32 run leaves its fp64 run by 1 % after 3 steps), so only the first
    decode steps carry a tight parity bar.
    family "tame": identical draws, except that the BatchNorm closing each residual branch has its weight scaled by
    TAME_LAST_BN_GAIN (the usual small-gamma residual initialisation).  Activations stay O(1-10) in both BN modes, the decoder
    works in its smooth regime, and the reference's fp32-vs-fp64 drift stays ~1e-5 of scale over all 16 steps -- so every step
    can be held to the north-star bar (1e-4 on logits, exact argmax).
    family "tame_sharp": "tame" with the two layers that emit the action logits (object_head.sal_layer_2 / sal_layer_3, weights and
    biases) scaled by SHARP_HEAD_GAIN: logits of trained-model magnitude (several units, peaked softmax) on a recurrence that stays
    non-chaotic -- the absolute 1e-4 bar on logits is then a RELATIVE 1e-5 (VERDICT r4 "what's weak" #4)."""
    if family not in FAMILIES:
        raise ValueError(f"unknown weight family {family!r}")
    if family == "tame_sharp":
        t = procedural_tensor(key, shape, bn_keys, seed, "tame")
        return t * SHARP_HEAD_GAIN if key.startswith(("object_head.sal_layer_2.", "object_head.sal_layer_3.")) else t
    rng = _rng(key, seed)
    prefix, _, leaf = key.rpartition(".")
    if leaf == "num_batches_tracked":
        return torch.zeros((), dtype=torch.int64)
    if leaf == "running_mean":
        return torch.from_numpy(rng.normal(0.0, 0.1, shape)).to(torch.float32)
    if leaf == "running_var":
        return torch.from_numpy(rng.uniform(0.5, 1.5, shape)).to(torch.float32)
    if prefix in bn_keys:
        if leaf == "weight":
            w = rng.uniform(0.5, 1.5, shape)
            if family == "tame" and _is_last_bn_of_block(prefix, bn_keys):
                w = w * TAME_LAST_BN_GAIN
            return torch.from_numpy(w).to(torch.float32)
        return torch.from_numpy(rng.normal(0.0, 0.1, shape)).to(torch.float32)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        gain = 2.0 if key.startswith("resnet.") else 1.0
        return torch.from_numpy(rng.normal(0.0, np.sqrt(gain / fan_in), shape)).to(torch.float32)
    return torch.from_numpy(rng.normal(0.0, 0.05, shape)).to(torch.float32)


def procedural_state_dict(spec: Mapping[str, Tuple[int, ...]], seed: int = 0, family: str = "default") -> Dict[str, torch.Tensor]:
    """``spec`` maps state_dict key -> shape (e.g. ``{k: tuple(v.shape) for k, v in model.state_dict().items()}``)."""
    bn = {k.rpartition(".")[0] for k in spec if k.endswith(".running_mean")}
    return {k: procedural_tensor(k, tuple(s), bn, seed, family) for k, s in spec.items()}


def fill_module(module: torch.nn.Module, seed: int = 0, family: str = "default") -> None:
    """Load procedural values into any module exposing the reference state_dict keys."""
    sd = module.state_dict()
    new = procedural_state_dict({k: tuple(v.shape) for k, v in sd.items()}, seed, family)
    module.load_state_dict({k: v.to(sd[k].dtype) for k, v in new.items()})
