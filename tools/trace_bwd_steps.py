"""Which side of the two-stream backward recurrence is the critical path, per decode step.

From a rocprofv3 --kernel-trace CSV of the bench step: for every launch of the h-gate conv's data gradient (halo build, side stream)
  dgrad   its start and duration
  chain   when the main stream's chain of small launches (backward of the memory update and of the previous step's heads) that was
          enqueued beside it ends = the end of the last main-stream launch before the fan-in that waits for the data gradient
          (sum_n_rows_kernel / sum_n_mixed_kernel)
  fan-in  when that fan-in starts, and the period to the next data gradient
and what the main stream ran meanwhile (kernel time by name).   python3 tools/trace_bwd_steps.py <trace.csv> [first_step_launch]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:60]
DG = "h2_kernel<1, 3, true, false, true"
dg = [i for i, r in enumerate(ev) if DG in r["Kernel_Name"]]
if not dg:
    sys.exit("no halo data-gradient launches in this trace")
# the launches of ONE training step: the last 17 of the trace's last step (15 h-gate + x-gate + sal_conv), or from argv[2]
first = int(sys.argv[2]) if len(sys.argv) > 2 else max(0, len(dg) - 17)
crit = collections.Counter()
between = collections.Counter()
tot_period = 0.0
print(f"{'#':>3} {'dgrad ms':>9} {'chain end':>10} {'dgrad end':>10} {'fan-in at':>10} {'period':>8}  critical")
for k in range(first, len(dg) - 1):
    a, b = dg[k], dg[k + 1]
    s0, e0 = int(ev[a]["Start_Timestamp"]), int(ev[a]["End_Timestamp"])
    q0 = ev[a].get("Queue_Id")
    fan = None
    chain_end = s0
    for r in ev[a + 1:b]:
        if r.get("Queue_Id") == q0:
            continue
        n = r["Kernel_Name"]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if fan is None and ("sum_n_rows_kernel" in n or "sum_n_mixed_kernel" in n) and s >= e0 - 2000:
            fan = s
        if fan is None:
            chain_end = max(chain_end, e)
            between[short(n)] += e - s
    if fan is None:
        continue
    period = (int(ev[b]["Start_Timestamp"]) - s0) / 1e6
    which = "dgrad" if e0 >= chain_end else "chain"
    crit[which] += 1
    tot_period += period
    print(f"{k:3d} {(e0 - s0) / 1e6:9.3f} {(chain_end - s0) / 1e6:10.3f} {(e0 - s0) / 1e6:10.3f} {(fan - s0) / 1e6:10.3f} {period:8.3f}  {which}")
print(f"critical path: {dict(crit)}; sum of periods {tot_period:.2f} ms")
print("main-stream kernel time beside the data gradients (ms):")
for n, t in between.most_common(25):
    print(f"  {t / 1e6:8.3f}  {n}")
