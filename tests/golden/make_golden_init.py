"""Per-parameter statistics (mean, std, abs-max, numel) of the REAL reference's freshly constructed models -- the initial
distributions of SURVEY.md §8 row a-13 (models/resnet.py:112-118 He-normal over k*k*Cout; mmcv xavier_init (normal) for the
decoder convs, normal_init(std=0.01) for the Linears, constant biases: AiR/models/baseline_attention.py:58-65,90-97,126-133,
176-185,495-504).  Data only (numbers); mmcv itself is absent here (mmcv==1.1.4, sp_baseline.yml:64), its three initialisers are
shimmed by their published definitions in make_golden.py, so the mmcv half stays "parity unpinned" by construction.

Usage: python tests/golden/make_golden_init.py   (build container only; writes init_stats.json next to this file)"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import load_reference  # noqa: E402

if __name__ == "__main__":
    out = {}
    for task in ("AiR", "OSIE", "COCO_Search18"):
        torch.manual_seed(0)
        model, _, _ = load_reference(task, "resnet50", 16)
        st = {}
        for k, p in model.named_parameters():
            v = p.detach().double()
            st[k] = [float(v.mean()), float(v.std()) if v.numel() > 1 else 0.0, float(v.abs().max()), int(v.numel())]
        out[task] = st
        print(task, len(st))
    with open(os.path.join(HERE, "init_stats.json"), "w") as f:
        json.dump(out, f)
