"""BatchNorm running statistics for the EVAL half of the bench-path parity case (tests/test_model_gpu.py::test_bench_path_at_320x512...):

    python tests/golden/make_bn_calibration.py          (build container or GPU box: the oracle only, no reference needed)

The procedural running statistics of the "tame" weight family (mean ~N(0, 0.1), var ~U(0.5, 1.5)) do not match the activations the
synthetic inputs produce at 320x512, so in eval mode the ORACLE's own fp32 run left its fp64 run after ~10 decode steps and rounds 3-4
could only compare "informative" steps (VERDICT r4 "what's weak" #3).  Trained checkpoints carry running statistics that fit their
data; this script gives the case the same: the batch statistics of its own two synthetic images, layer by layer, from ONE train-mode pass
of the fp64 oracle's encoder (momentum undone: batch = (new - 0.9 old) / 0.1; the variance is torch's unbiased running estimate).
Stored: tests/golden/bench_bn_calib.npz = {"<bn>.running_mean", "<bn>.running_var"} for the 53 BatchNorm layers (float64, 420 KB).
Both the oracle and the HIP model load them for the eval forward; train mode ignores running statistics."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from helpers import oracle_state          # noqa: E402
from oracle import scanpath_oracle as O   # noqa: E402
from scanpaths_amd.synth import make_batch  # noqa: E402

SEED, NB, H, W, T = 21, 2, 320, 512, 16      # = the bench-path case


def main():
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    b = make_batch("AiR", NB, H, W, T, seed=SEED)
    sd = oracle_state("AiR", "resnet50", SEED, H // 8, W // 8, dtype=torch.float64, family="tame")
    bn_new = {}
    with torch.no_grad():
        O.encoder(sd, b["images"].double(), "resnet50", training=True, bn_new=bn_new)
    out = {}
    for k, v in bn_new.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            out[k] = ((v - 0.9 * sd[k]) / 0.1).numpy()
    assert len(out) == 2 * 53, len(out)
    np.savez_compressed(os.path.join(HERE, "bench_bn_calib.npz"), **out)
    print(len(out), "vectors,", sum(v.size for v in out.values()), "values; e.g. resnet.1.running_var[:4] =", out["resnet.1.running_var"][:4])


if __name__ == "__main__":
    main()
