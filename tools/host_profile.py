#!/usr/bin/env python3
"""Host side of one training step (bs 32, 320x512, T = 16): wall time to ENQUEUE forward / loss / backward / optimiser onto an idle device,
and a cProfile of three steps sorted by own time (where the Python side of ~1 900 launches per step spends its time)."""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from scanpaths_amd.models.baseline_attention import baseline
    from scanpaths_amd.models.loss import supervised_loss
    from scanpaths_amd.optim import FlatAdam
    from scanpaths_amd.procedural import fill_module
    from scanpaths_amd.synth import make_batch
    dev = torch.device("cuda", 0)
    T = 16
    model = baseline(convLSTM_length=T, map_width=64, map_height=40)
    fill_module(model, seed=0)
    model = model.to(dev).train()
    b = {k: v.to(dev) for k, v in make_batch("AiR", 32, 320, 512, T, seed=0).items()}
    opt = FlatAdam(model.parameters(), lr=1e-4, weight_decay=5e-5, clip=12.5)
    marks = {}

    def step(record=False):
        t = [time.perf_counter()]
        opt.zero_grad()
        pred = model(b["images"], b["attention_maps"], b["performances"])
        t.append(time.perf_counter())
        loss, _, _ = supervised_loss(pred, b["scanpaths"], b["durations"], b["action_masks"], b["duration_masks"], 1.0)
        t.append(time.perf_counter())
        loss.backward()
        t.append(time.perf_counter())
        opt.step()
        t.append(time.perf_counter())
        if record:
            for n, a, c in zip(("forward", "loss", "backward", "optimiser"), t, t[1:]):
                marks.setdefault(n, []).append((c - a) * 1e3)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    for _ in range(3):
        step(True)
        t0 = time.perf_counter()
        torch.cuda.synchronize()
        marks.setdefault("device still busy after the host is done", []).append((time.perf_counter() - t0) * 1e3)
    # the same without a synchronise between the steps: does the host run ahead of the device, and if not, where is it held?
    marks2 = {}
    for i in range(8):
        t = [time.perf_counter()]
        opt.zero_grad()
        t.append(time.perf_counter())
        pred = model(b["images"], b["attention_maps"], b["performances"])
        t.append(time.perf_counter())
        loss, _, _ = supervised_loss(pred, b["scanpaths"], b["durations"], b["action_masks"], b["duration_masks"], 1.0)
        loss.backward()
        t.append(time.perf_counter())
        opt.step()
        t.append(time.perf_counter())
        marks2[i] = [round((c - a) * 1e3, 1) for a, c in zip(t, t[1:])]
    t0 = time.perf_counter()
    torch.cuda.synchronize()
    tail = (time.perf_counter() - t0) * 1e3
    print("8 steps back to back, no synchronise: host ms in [zero_grad, forward, loss + backward, optimiser] per step")
    for i, v in marks2.items():
        print("  step", i, v)
    print(f"  device still busy after the host is done with step 7: {tail:.1f} ms")
    print("host enqueue time onto an idle device, ms (3 steps):")
    for n, v in marks.items():
        print(f"  {n:45s} {min(v):8.2f} .. {max(v):8.2f}")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        step()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
    print(s.getvalue()[:9000])


if __name__ == "__main__":
    main()
