"""ScanMatch scoring throughput: the batched HIP Needleman-Wunsch against the oracle (numpy/python
restatement of the reference's DP loop) on the validation-shaped workload -- 10 sampled scanpaths per image scored against
the image's human scanpaths, evaluation configuration (320x240, 16x12 bins, TempBin 50, Threshold 3.5).
    python tests/diagnostics/bench_scanmatch.py [--images 2000] [--humans 6] [--samples 10] [--cpu-pairs 40]
Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=2000)
    ap.add_argument("--humans", type=int, default=6)
    ap.add_argument("--samples", type=int, default=10)
    ap.add_argument("--cpu-pairs", type=int, default=40)
    a = ap.parse_args()
    from scanpaths_amd.utils.evaltools.scanmatch import ScanMatch
    from oracle import scanmatch_oracle as SO
    g = np.random.Generator(np.random.PCG64(0))

    def scanpath():
        L = int(g.integers(2, 17))
        return np.stack([g.uniform(0, 320, L), g.uniform(0, 240, L), g.uniform(80, 600, L)], 1)

    hum = [scanpath() for _ in range(a.images * a.humans)]
    smp = [scanpath() for _ in range(a.images * a.samples)]
    pairs = torch.tensor([(i * a.humans + h, i * a.samples + s) for i in range(a.images) for h in range(a.humans)
                          for s in range(a.samples)], dtype=torch.int32)
    sm = ScanMatch(Xres=320, Yres=240, Xbin=16, Ybin=12, Offset=(0, 0), TempBin=50, Threshold=3.5)
    sh, lh = sm.sequences(hum)
    ss, ls = sm.sequences(smp)
    pd = pairs.to(sm.device)
    sm.match_pairs(sh, lh, ss, ls, pd)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        scores = sm.match_pairs(sh, lh, ss, ls, pd)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    cells = float((lh[pairs[:, 0].long()].double() * ls[pairs[:, 1].long()].double()).sum().item())
    # end to end incl. host -> device of the fixations and the sequence kernel
    t0 = time.perf_counter()
    sh2, lh2 = sm.sequences(hum)
    ss2, ls2 = sm.sequences(smp)
    sc2 = sm.match_pairs(sh2, lh2, ss2, ls2, pd).cpu()
    t_e2e = time.perf_counter() - t0
    # CPU: the oracle's DP (same loop structure as the reference's pure-python match()) on a bounded sample
    S = SO.submatrix(16, 12, 3.5)
    idx = g.choice(len(pairs), a.cpu_pairs, replace=False)
    shc, lhc, ssc, lsc = sh.cpu().numpy(), lh.cpu().numpy(), ss.cpu().numpy(), ls.cpu().numpy()
    t0 = time.perf_counter()
    ccells = 0
    for k in idx:
        i, j = pairs[k].tolist()
        v = SO.nw_score(shc[i, :lhc[i]], ssc[j, :lsc[j]], S, 0.0)
        assert v == float(scores[k].item()) or (np.isnan(v) and np.isnan(float(scores[k].item())))
        ccells += int(lhc[i]) * int(lsc[j])
    t_cpu = time.perf_counter() - t0
    print(json.dumps({
        "metric": "ScanMatch pairs/s (validation-shaped: images x humans x samples, TempBin 50)",
        "pairs": len(pairs), "mean_len": float(torch.cat([lh, ls]).double().mean().item()),
        "kernel_ms": ms, "pairs_per_s": len(pairs) / ms * 1e3, "dp_cells_per_s": cells / ms * 1e3,
        "end_to_end_s": t_e2e, "end_to_end_pairs_per_s": len(pairs) / t_e2e,
        "cpu_oracle": {"pairs": int(a.cpu_pairs), "seconds": t_cpu, "pairs_per_s": a.cpu_pairs / t_cpu,
                       "dp_cells_per_s": ccells / t_cpu, "cores": 1, "kind": "port", "checked_bit_exact": True},
    }))


if __name__ == "__main__":
    main()
