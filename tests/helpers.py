"""Shared helpers for the parity tests (oracle <-> golden <-> HIP)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        meta = json.load(f)
    data = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    return meta, data


def oracle_state(task, arch, seed, Hm=30, Wm=40, dtype=torch.float64, family="default"):
    from scanpaths_amd.procedural import procedural_state_dict
    from scanpaths_amd.spec import model_spec
    sd = procedural_state_dict(model_spec(task, arch, Hm, Wm), seed, family)
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}


def case_inputs(meta, dtype=torch.float64):
    from scanpaths_amd.synth import make_batch
    b = make_batch(meta["task"], meta["B"], meta["H"], meta["W"], meta["T"], seed=meta["seed"])
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in b.items()}


def max_err(a, b):
    if a is None:
        a = torch.zeros_like(torch.as_tensor(b))
    a = torch.as_tensor(a).detach().cpu().to(torch.float64)
    b = torch.as_tensor(b).detach().cpu().to(torch.float64)
    return (a - b).abs().max().item()
