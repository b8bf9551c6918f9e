#!/usr/bin/env python3
"""End-to-end walk through the path on synthetic data, written the way the reference's AiR/train.py drives its model
(supervised phase :176-211, RL phase :212-345, validation sampling + ScanMatch scoring :347-372 / utils/evaluation.py):

    python examples/train_synthetic.py [--iters 3] [--rl-iters 1] [--batch 4]

Everything numeric runs in the HIP kernels of scanpaths_amd; the script only moves tensors and prints scalars."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scanpaths_amd.models.baseline_attention import baseline                      # noqa: E402
from scanpaths_amd.models.loss import CrossEntropyLoss, MLPLogNormalDistribution  # noqa: E402
from scanpaths_amd.models.sampling import Sampling                                # noqa: E402
from scanpaths_amd.optim import FlatAdam                                          # noqa: E402
from scanpaths_amd.opts import parse_opt                                          # noqa: E402
from scanpaths_amd.procedural import fill_module                                  # noqa: E402
from scanpaths_amd.rl import rl_step, rl_step_single_head                         # noqa: E402
from scanpaths_amd.synth import make_batch                                        # noqa: E402
from scanpaths_amd.utils.evaltools.scanmatch import ScanMatch                     # noqa: E402

DT = {"names": ("start_x", "start_y", "duration"), "formats": ("f8", "f8", "f8")}


def human_scanpaths(g, n_images):
    """synthetic 'ground-truth' fixation vectors: per image 2 good- and 2 poor-performance scanpaths (seconds)"""
    gt, perf = [], []
    for _ in range(n_images):
        paths = []
        for _ in range(4):
            L = int(g.integers(3, 9))
            fv = np.zeros(L, dtype=DT)
            fv["start_x"], fv["start_y"], fv["duration"] = g.uniform(0, 320, L), g.uniform(0, 240, L), g.uniform(0.1, 0.6, L)
            paths.append(fv)
        gt.append(paths)
        perf.append([True, True, False, False])
    return gt, perf


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--rl-iters", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4)
    a = ap.parse_args()
    args = parse_opt("AiR", [])                                 # the reference's flags and defaults (opts.py)
    dev = torch.device("cuda:0")
    T = 8
    model = baseline(convLSTM_length=T, min_length=args.min_length, map_width=args.map_width, map_height=args.map_height)
    fill_module(model, seed=0)
    model = model.to(dev)
    optimizer = FlatAdam(model.parameters(), lr=args.lr, weight_decay=args.weight_decay, clip=args.clip,
                         conditional_params=getattr(model, "has_conditional_params", False),
                         reference_zero_grad=getattr(model, "has_conditional_params", False))   # torch-1.6 zero_grad semantics for COCO heads
    sampling = Sampling(convLSTM_length=T, min_length=args.min_length, map_width=args.map_width, map_height=args.map_height,
                        width=args.width, height=args.height)
    g = np.random.Generator(np.random.PCG64(0))

    # ---- supervised phase (AiR/train.py:176-211) ----
    model.train()
    for it in range(a.iters):
        batch = {k: v.to(dev) for k, v in make_batch("AiR", a.batch, args.height, args.width, T, seed=it).items()}
        optimizer.zero_grad()
        predicts = model(batch["images"], batch["attention_maps"], batch["performances"])
        loss_actions = CrossEntropyLoss(predicts["all_actions_prob"], batch["scanpaths"], batch["action_masks"])
        loss_duration = MLPLogNormalDistribution(predicts["log_normal_mu"], predicts["log_normal_sigma2"], batch["durations"],
                                                 batch["duration_masks"])
        loss = loss_actions + args.lambda_1 * loss_duration
        loss.backward()
        grad_norm = optimizer.step()
        print(f"supervised iter {it}: loss {float(loss):.4f} (actions {float(loss_actions):.4f}, duration {float(loss_duration):.4f}) "
              f"grad norm {float(grad_norm):.2f}")

    # ---- RL phase (AiR/train.py:212-345) ----
    # (with random weights and barely-updated BatchNorm running statistics the eval-mode activations are huge and the gates
    #  saturate, so the raw gradient norm printed below is enormous; FlatAdam clips it to args.clip as the reference does)
    cfg = dict(Xres=args.width, Yres=args.height, Xbin=16, Ybin=12, Offset=(0, 0), Threshold=3.5)
    sm_wd, sm_wod = ScanMatch(TempBin=50, **cfg), ScanMatch(**cfg)
    for it in range(a.rl_iters):
        batch = {k: v.to(dev) for k, v in make_batch("AiR", a.batch, args.height, args.width, T, seed=100 + it).items()}
        gt, perf = human_scanpaths(g, a.batch)
        loss, info = rl_step(model, sampling, optimizer, batch["images"], batch["attention_maps"], gt, perf, sm_wd, sm_wod,
                             rl_sample_number=2, lambda_5=args.lambda_5)
        print(f"rl iter {it}: loss {float(loss):.5f}  mean same-performance reward {float(np.mean(info['same_reward_hmean'])):.4f} "
              f"grad norm {float(info['grad_norm']):.3f}  resamples {info['resamples']}")

    # ---- validation-style sampling + scoring (AiR/train.py:347-372, utils/evaluation.py) ----
    model.eval()
    batch = {k: v.to(dev) for k, v in make_batch("AiR", a.batch, args.height, args.width, T, seed=999).items()}
    with torch.no_grad():
        predict = model(batch["images"], batch["attention_maps"])
    gt, _ = human_scanpaths(g, a.batch)
    rows = []
    for rep in range(3):
        s = sampling.random_sample(predict["good_all_actions_prob"], predict["good_log_normal_mu"], predict["good_log_normal_sigma2"])
        fix, _, _ = sampling.generate_scanpath(batch["images"], s["selected_actions_probs"], s["durations"], s["selected_actions"])
        ms = lambda fv: np.stack([fv["start_x"], fv["start_y"], fv["duration"] * 1000], 1)
        for i in range(a.batch):
            rows.append(sm_wd.match_all([ms(f) for f in gt[i]], [ms(fix[i])]).mean())
    print(f"validation: mean ScanMatch (with duration) of sampled vs human scanpaths {float(np.nanmean(rows)):.4f} over {len(rows)} samples")

    # ---- the single-head tasks: one OSIE RL iteration and its validation table (OSIE/train.py:197-262, OSIE/utils/evaluation.py:151-282) ----
    from scanpaths_amd.models.baseline_attention import baseline_osie
    from scanpaths_amd.utils.evaluation_osie import evaluation
    osie = baseline_osie(convLSTM_length=T, arch="resnet18")
    fill_module(osie, seed=1)
    osie = osie.to(dev)
    opt2 = FlatAdam(osie.parameters(), lr=args.lr, weight_decay=5e-4, clip=args.clip)
    samp2 = Sampling(convLSTM_length=T, min_length=3, map_width=args.map_width, map_height=args.map_height, width=args.width, height=args.height)
    images = make_batch("OSIE", a.batch, args.height, args.width, T, seed=7)["images"].to(dev)
    gt, _ = human_scanpaths(g, a.batch)
    loss, info = rl_step_single_head(osie, samp2, opt2, images, gt, sm_wd, sm_wod, "OSIE", rl_sample_number=2)
    print(f"OSIE rl iter: loss {float(loss):.5f}  reward hmean {float(info['reward_hmean'].mean()):.4f}  grad norm {float(info['grad_norm']):.3f}")
    with torch.no_grad():
        p = osie(images)
    s = samp2.random_sample(p["all_actions_prob"], p["log_normal_mu"], p["log_normal_sigma2"])
    fix, _, _ = samp2.generate_scanpath(images, s["selected_actions_probs"], s["durations"], s["selected_actions"])
    cur, _, _ = evaluation(gt, fix)
    print("OSIE validation:", {k: {m: round(float(v), 4) for m, v in d.items()} for k, d in cur.items()})


if __name__ == "__main__":
    main()
