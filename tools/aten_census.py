#!/usr/bin/env python3
"""Which lines of this package make PyTorch launch its own kernels inside the training step (copies, adds, cats, fills)?

One bench step (bs 32, 320x512, T = 16) under torch.profiler with Python stacks; every aten operator that launched a device kernel
is attributed to the innermost frame inside scanpaths_amd/, or -- where the build records no Python stacks -- to the chain of enclosing
operators and autograd nodes (bwd:XBackward = the backward of X; a bare add_ under no node = gradient accumulation of a tensor with several consumers).  Prints launches per step by (operator, site).

    python3 tools/aten_census.py [--top 60]
"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=60)
    args = ap.parse_args()
    from scanpaths_amd.models.baseline_attention import baseline
    from scanpaths_amd.models.loss import supervised_loss
    from scanpaths_amd.optim import FlatAdam
    from scanpaths_amd.procedural import fill_module
    from scanpaths_amd.synth import make_batch
    from torch.profiler import ProfilerActivity, profile
    dev = torch.device("cuda", 0)
    T = 16
    model = baseline(convLSTM_length=T, map_width=64, map_height=40)
    fill_module(model, seed=0)
    model = model.to(dev).train()
    b = {k: v.to(dev) for k, v in make_batch("AiR", 32, 320, 512, T, seed=0).items()}
    opt = FlatAdam(model.parameters(), lr=1e-4, weight_decay=5e-5, clip=12.5)

    def step():
        opt.zero_grad()
        pred = model(b["images"], b["attention_maps"], b["performances"])
        loss, _, _ = supervised_loss(pred, b["scanpaths"], b["durations"], b["action_masks"], b["duration_masks"], 1.0)
        loss.backward()
        opt.step()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    count = collections.Counter()
    dev_us = collections.Counter()
    for e in prof.events():
        if not e.name.startswith("aten::") or not e.kernels:
            continue
        if any(c.name.startswith("aten::") and c.kernels for c in e.cpu_children):
            continue          # count the innermost operator that launched
        site = None
        for fr in (e.stack or []):
            if "scanpaths_amd/" in fr:
                site = fr.split("scanpaths_amd/")[-1]
                break
        if site is None:          # no Python stack on this build: the chain of enclosing operators / autograd nodes instead
            chain, q = [], e.cpu_parent
            while q is not None and len(chain) < 4:
                if not q.name.startswith("aten::") or not chain:
                    chain.append(q.name.replace("autograd::engine::evaluate_function: ", "bwd:"))
                q = q.cpu_parent
            site = " < ".join(chain) if chain else "top level"
        count[(e.name, site)] += len(e.kernels)
        dev_us[(e.name, site)] += sum(k.duration for k in e.kernels)
    tot = sum(count.values())
    print(f"{tot} device launches by aten operators in one step")
    for (name, site), n in count.most_common(args.top):
        print(f"{n:5d}  {dev_us[(name, site)]:8.0f} us  {name:28s} {site[:150]}")


if __name__ == "__main__":
    main()
