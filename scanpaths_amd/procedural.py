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


def procedural_tensor(key: str, shape: Tuple[int, ...], bn_keys: Iterable[str], seed: int = 0) -> torch.Tensor:
    """Value for one state_dict entry. ``bn_keys`` = set of module prefixes that are BatchNorm."""
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
            return torch.from_numpy(rng.uniform(0.5, 1.5, shape)).to(torch.float32)
        return torch.from_numpy(rng.normal(0.0, 0.1, shape)).to(torch.float32)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        gain = 2.0 if key.startswith("resnet.") else 1.0
        return torch.from_numpy(rng.normal(0.0, np.sqrt(gain / fan_in), shape)).to(torch.float32)
    return torch.from_numpy(rng.normal(0.0, 0.05, shape)).to(torch.float32)


def procedural_state_dict(spec: Mapping[str, Tuple[int, ...]], seed: int = 0) -> Dict[str, torch.Tensor]:
    """``spec`` maps state_dict key -> shape (e.g. ``{k: tuple(v.shape) for k, v in model.state_dict().items()}``)."""
    bn = {k.rpartition(".")[0] for k in spec if k.endswith(".running_mean")}
    return {k: procedural_tensor(k, tuple(s), bn, seed) for k, s in spec.items()}


def fill_module(module: torch.nn.Module, seed: int = 0) -> None:
    """Load procedural values into any module exposing the reference state_dict keys."""
    sd = module.state_dict()
    new = procedural_state_dict({k: tuple(v.shape) for k, v in sd.items()}, seed)
    module.load_state_dict({k: v.to(sd[k].dtype) for k, v in new.items()})
