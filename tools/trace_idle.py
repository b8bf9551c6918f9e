"""Where is the GPU EMPTY inside one traced training step?  (rocprofv3 --kernel-trace CSV of bench.py)

Union of the kernels' [start, end] intervals over both streams of one step; the time covered by no kernel is split by phase
(forward = up to the loss kernel, backward recurrence = up to the deferred h-gate weight gradient, rest) and the largest empty intervals
are listed with the kernel that ends before and the one that starts after them.  Empty time between dependent tiny kernels is launch
latency; empty time that grows with the number of launches per millisecond is the host falling behind.
    python3 tools/trace_idle.py <..._kernel_trace.csv> [step index, default: the last complete one]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")) for r in rows))
steps = [i for i, e in enumerate(ev) if e[2].startswith("clip_adam_kernel")]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(steps) - 2
seg = ev[steps[k] + 1: steps[k + 1] + 1]
t0, t1 = seg[0][0], seg[-1][1]


def first(prefix, after=0):
    for i, e in enumerate(seg):
        if i >= after and e[2].startswith(prefix):
            return i
    return None


i_loss = first("loss_rows_kernel") or first("loss_")
i_hw2 = max((i for i, e in enumerate(seg) if e[2].startswith("hw2_kernel")), key=lambda i: seg[i][1] - seg[i][0], default=None)
marks = [("forward", t0, seg[i_loss][0] if i_loss is not None else t1)]
if i_loss is not None and i_hw2 is not None:
    marks += [("backward: decoder recurrence (up to the deferred h-gate weight gradient)", seg[i_loss][0], seg[i_hw2][0]),
              ("backward: deferred weight gradient + encoder + optimiser", seg[i_hw2][0], t1)]
gaps = []
cur_end, last_name = seg[0][1], seg[0][2]
for s, e, n in seg[1:]:
    if s > cur_end:
        gaps.append((cur_end, s, last_name, n))
    if e > cur_end:
        cur_end, last_name = e, n
print(f"step {k}: {len(seg)} launches, wall {(t1 - t0) / 1e6:.2f} ms, empty {sum(b - a for a, b, _, _ in gaps) / 1e6:.2f} ms in {len(gaps)} intervals")
for name, a, b in marks:
    g = [(min(y, b) - max(x, a)) for x, y, _, _ in gaps if y > a and x < b]
    n_l = sum(1 for s, _, _ in seg if a <= s < b)
    print(f"  {name}: {(b - a) / 1e6:7.2f} ms, {n_l} launches ({n_l / max((b - a) / 1e6, 1e-9):.0f} per ms), empty {sum(g) / 1e6:6.2f} ms "
          f"in {len(g)} intervals (median {sorted(g)[len(g) // 2] / 1e3 if g else 0:.1f} us)")
print("largest empty intervals:")
for a, b, p, n in sorted(gaps, key=lambda g: g[0] - g[1])[:25]:
    print(f"  {(b - a) / 1e3:7.1f} us at {(a - t0) / 1e6:7.2f} ms   after {p.split('(')[0][:48]:48s} before {n.split('(')[0][:48]}")
