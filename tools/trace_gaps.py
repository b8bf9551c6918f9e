"""Time line of ONE traced training step from a rocprofv3 --kernel-trace CSV: where the GPU is inside the three dominant GEMM families of
the decoder and what runs between consecutive launches of them (the small kernels of the decode loop, which cannot fill the chip).
    python3 tools/trace_gaps.py <..._kernel_trace.csv> [step index, default: the last complete one]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "")) for r in rows))
steps = [i for i, e in enumerate(ev) if e[2].startswith("clip_adam_kernel")]          # one per training step
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(steps) - 2
seg = ev[steps[k] + 1: steps[k + 1] + 1]
t0, t1 = seg[0][0], seg[-1][1]
busy = sum(e - s for s, e, _ in seg)
print(f"step {k}: {len(seg)} launches, wall {(t1 - t0) / 1e6:.2f} ms, sum of kernel durations {busy / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms")


def fam(n):
    if n.startswith("void h2_kernel<0, 3, true, true"):
        return "fused fwd"
    if n.startswith("void h2_kernel<1, 3, true, false, true"):
        return "halo dgrad"
    if n.startswith("hw2_kernel"):
        return "hw2"
    return None


for name in ("fused fwd", "halo dgrad"):
    idx = [i for i, e in enumerate(seg) if fam(e[2]) == name]
    gaps, inside = [], collections.Counter()
    for a, b in zip(idx, idx[1:]):
        gaps.append((seg[b][0] - seg[a][1]) / 1e3)
        for s, e, n in seg[a + 1:b]:
            inside[n.split("(")[0].replace("void ", "")[:60]] += (e - s) / 1e3
    if not gaps:
        continue
    print(f"\\n{name}: {len(idx)} launches, {sum((seg[i][1] - seg[i][0]) for i in idx) / 1e6:.2f} ms inside; between consecutive launches: "
          f"mean {sum(gaps) / len(gaps):.0f} us, min {min(gaps):.0f}, max {max(gaps):.0f}, total {sum(gaps) / 1e3:.2f} ms")
    tot = sum(inside.values())
    print(f"  kernels between them: {tot / 1e3:.2f} ms of kernel time ({100 * tot / max(sum(gaps), 1e-9):.0f} % of the gaps); largest:")
    for n, v in inside.most_common(14):
        print(f"    {v / 1e3:7.2f} ms  {n}")
