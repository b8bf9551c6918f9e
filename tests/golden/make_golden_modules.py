"""Per-module goldens from the REAL reference modules (SURVEY.md §8c fixture plan (iv)).

Imports chenxy99/Scanpaths where it lies (/root/reference, build container only -- the reference never travels), builds ONE module
at a time with the reference's own constructors
  ConvLSTM.forward                   AiR/models/baseline_attention.py:37-56       (embed 256, 16x16 map: the fused-cell kernel's shape class)
  nn.Conv2d(512,512,5) -> predict_head   :306-309 + :149-174 (train and eval)     (30x40 map: the reference hard-codes it, :142,145)
  spatial_att.forward                :111-124
  semantic_att.forward               :77-88
  Bottleneck.forward                 AiR/models/resnet.py:57-93, dilated like baseline_attention.py:226-238: a layer-3 block
                                     (dilation 2) and a layer2[0]-type block (stride forced to 1, with downsample), train AND eval
fills it with scanpaths_amd.procedural values under the MODEL's state_dict key names (so the oracle's functions and the HIP ops are fed
the same numbers from the same keys), runs it in fp64 and fp32 on seeded inputs (scanpaths_amd-independent numpy PCG64 draws, regenerated
by the tests from the stored seeds), and stores outputs, input gradients and parameter-gradient norms as DATA in modules.npz.

Usage:  python tests/golden/make_golden_modules.py
"""
from __future__ import annotations

import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
from make_golden import REF, _install_shims  # noqa: E402
from scanpaths_amd.procedural import procedural_state_dict  # noqa: E402

sys.path.insert(0, os.path.join(REPO, "tests"))
from module_cases import MODULE_CASES, case_inputs, cotangents  # noqa: E402  (shapes + seeds shared with the tests)


def _ref_modules():
    _install_shims()
    for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
        del sys.modules[k]
    sys.path[:] = [p for p in sys.path if not p.startswith(REF)]
    sys.path.insert(0, os.path.join(REF, "AiR"))
    return importlib.import_module("models.baseline_attention"), importlib.import_module("models.resnet")


def _load(mod: torch.nn.Module, prefix: str, seed: int, dt):
    """procedural values under the model-level key names `prefix + <module key>`"""
    sd = mod.state_dict()
    vals = procedural_state_dict({prefix + k: tuple(v.shape) for k, v in sd.items()}, seed)
    mod.load_state_dict({k: vals[prefix + k].to(sd[k].dtype) for k in sd})
    return mod.to(dt)


def _grads(name, outs, inputs, params, tag, store):
    """scalar = sum of outputs weighted by the fixed seeded cotangents of module_cases.cotangents; stores the input gradients in full,
    parameter gradients in full up to 64k elements and as norms beyond"""
    cots = cotangents(name, outs)
    tot = 0.0
    for k, o in outs.items():
        tot = tot + (o * cots[k].to(o.dtype)).sum()
    gi = torch.autograd.grad(tot, list(inputs.values()) + list(params.values()), allow_unused=True)
    for (k, _), gr in zip(inputs.items(), gi[:len(inputs)]):
        if gr is None:
            continue
        if gr.numel() > (1 << 19):          # the head's input gradient [B,512,30,40]: every 8th channel in full + the norm of the whole
            store[f"{tag}/d_{k}_sub8"] = gr.detach()[:, ::8].contiguous().numpy()
            store[f"{tag}/dnorm_in/{k}"] = np.array(float(gr.norm()))
        else:
            store[f"{tag}/d_{k}"] = gr.detach().numpy()
    for (k, p), gr in zip(params.items(), gi[len(inputs):]):
        if gr is None:
            continue
        store[f"{tag}/dnorm/{k}"] = np.array(float(gr.norm()))
        if gr.numel() <= 65536:
            store[f"{tag}/dparam/{k}"] = gr.detach().numpy()


def main():
    M, R = _ref_modules()
    store = {}
    for name, cs in MODULE_CASES.items():
        for dt, tag in ((torch.float64, "ref64"), (torch.float32, "ref32")):
            T = f"{name}/{tag}"
            ins = {k: v.to(dt).requires_grad_(v.is_floating_point()) for k, v in case_inputs(name).items()}
            seed = cs["seed"]
            if cs["kind"] == "convlstm":
                mod = _load(M.ConvLSTM(embed_size=cs["C"]), "lstm.", seed, dt)
                h2, (h2b, c2) = mod(ins["x"], (ins["h"], ins["c"]), ins["sp_pos"], ins["sp_neg"], ins["se_pos"], ins["se_neg"])
                assert h2 is h2b
                outs = {"h": h2, "c": c2}
                params = {"lstm." + k: p for k, p in mod.named_parameters()}
            elif cs["kind"] == "head":
                conv = _load(torch.nn.Conv2d(512, 512, kernel_size=5, padding=2, stride=1, bias=True), "performance_sal_layer.True.", seed, dt)
                head = _load(M.predict_head(16), "object_head.", seed, dt)
                head.train(cs["training"])
                o = head(conv(ins["h"]))
                outs = {"actions": o["actions"], "log_normal_mu": o["log_normal_mu"], "log_normal_sigma2": o["log_normal_sigma2"],
                        "action_map": o["action_map"]}
                params = {"performance_sal_layer.True." + k: p for k, p in conv.named_parameters()}
                params.update({"object_head." + k: p for k, p in head.named_parameters()})
            elif cs["kind"] == "spatial_att":
                mod = _load(M.spatial_att(cs["W"], cs["H"]), "spatial_att.", seed, dt)
                outs = {"mem": mod(ins["lists"], ins["cur"])}
                params = {"spatial_att." + k: p for k, p in mod.named_parameters()}
            elif cs["kind"] == "semantic_att":
                mod = _load(M.semantic_att(embed_size=cs["E"]), "semantic_att.", seed, dt)
                outs = {"mem": mod(ins["lists"], ins["cur"])}
                params = {"semantic_att." + k: p for k, p in mod.named_parameters()}
            elif cs["kind"] == "bottleneck":
                down = None
                if cs["down"]:
                    down = torch.nn.Sequential(torch.nn.Conv2d(cs["inpl"], cs["planes"] * 4, kernel_size=1, stride=cs["stride"], bias=False),
                                               torch.nn.BatchNorm2d(cs["planes"] * 4))
                mod = R.Bottleneck(cs["inpl"], cs["planes"], cs["stride"], down)
                mod.conv2.dilation, mod.conv2.padding = (cs["dil"], cs["dil"]), (cs["dil"], cs["dil"])      # dilate_resnet, :232-238
                mod = _load(mod, cs["prefix"], seed, dt)
                mod.train(cs["training"])
                outs = {"y": mod(ins["x"])}
                params = {cs["prefix"] + k: p for k, p in mod.named_parameters()}
                if cs["training"]:
                    for k, v in mod.state_dict().items():
                        if "running_" in k:
                            store[f"{T}/bn_after/{cs['prefix']}{k}"] = v.detach().numpy().copy()
            else:
                raise KeyError(cs["kind"])
            for k, v in outs.items():
                store[f"{T}/{k}"] = v.detach().numpy()
            _grads(name, outs, {k: v for k, v in ins.items() if v.requires_grad}, params, T, store)
    # What is committed: the fp64 run in full (outputs as float64; gradients rounded to float32 -- 6e-8 relative, far below the 2e-6
    # .. 1e-5 bars they are compared at) and, per array, how far the reference's OWN fp32 run lies from it ("err32/...": max |ref32 -
    # ref64|, the noise floor the GPU tests scale their bars with).
    final = {}
    for k, v in store.items():
        name, tag, rest = k.split("/", 2)
        if tag != "ref64":
            continue
        v32 = store[f"{name}/ref32/{rest}"]
        final[f"{name}/err32/{rest}"] = np.array(float(np.max(np.abs(v32.astype(np.float64) - v))) if v.size else 0.0)
        final[k] = v.astype(np.float32) if (rest.startswith("d_") or rest.startswith("dparam/")) else v
    store = final
    path = os.path.join(HERE, "modules.npz")
    np.savez_compressed(path, **store)
    print(f"wrote {path}: {len(store)} arrays, {os.path.getsize(path) / 2 ** 20:.1f} MiB")


if __name__ == "__main__":
    main()
