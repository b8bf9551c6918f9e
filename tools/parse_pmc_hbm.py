"""HBM-side traffic of the HBM-bound kernels of one bench.py run (north_star: "evidenced by rocprof HBM GB/s ... counters").
   python tools/parse_pmc_hbm.py <dir with fetch/ write/ trace/ sub-dirs> <out.json>
fetch/, write/: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes, p_counter_collection.csv); trace/: --kernel-trace
(p_kernel_trace.csv) of the SAME command.  Per kernel name: launches, average duration, FETCH_SIZE (doubled: gfx950 tallies the
128-byte requests of 16 B/lane streams at 64 B, /opt/skills/guides/MI355X_MICROARCH.md section HBM) and WRITE_SIZE in bytes per
launch, and (2*FETCH + WRITE) / duration against the 8 TB/s HBM3E peak.  Infinity-Cache hits are included in FETCH_SIZE."""
import collections
import csv
import glob
import json
import os
import sys

root, out = sys.argv[1], sys.argv[2]
KEEP = ("bn_apply_split_kernel", "bn_bwd_apply_split_kernel", "bn_bwd_mm_partial", "lstm_bwd_kernel", "sum_n_kernel", "sum_n_rows_kernel", "sum_n_mixed_kernel",
        "clip_adam_kernel", "bn_apply_kernel", "split2_kernel", "split2_cols_kernel", "sempool_bwd_kernel", "sempool_fwd_kernel", "lstm_rank1_fwd_kernel",
        "listatt_fwd_kernel", "listatt_bwd_kernel", "drt_fwd_kernel", "drt_bwd_data_kernel", "drt_bwd_weight_kernel", "drt_slab_reduce_kernel",
        "drt_batched", "sal_gather_fwd_kernel", "sal_gather_bwd_kernel", "skinny_kernel", "skinny_reduce_kernel", "head_fwd_kernel", "head_bwd_kernel",
        "head_dur", "maxpool", "rank1_")


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def find(sub, pat):
    fs = glob.glob(os.path.join(root, sub, "**", pat), recursive=True)
    return fs[0] if fs else None


ctr = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("fetch", "write"):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        n = short(r["Kernel_Name"])
        if n.startswith(KEEP):
            ctr[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
f = find("trace", "*kernel_trace.csv")
if f:
    for r in csv.DictReader(open(f)):
        n = short(r["Kernel_Name"])
        if n.startswith(KEEP):
            dur[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)       # us
dur_beside = collections.defaultdict(list)
f2 = find("trace_async", "*kernel_trace.csv")
if f2:
    for r in csv.DictReader(open(f2)):
        n = short(r["Kernel_Name"])
        if n.startswith(KEEP):
            dur_beside[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
res = {}
for n in sorted(set(ctr) | set(dur)):
    d = {"launches": len(dur.get(n, []))}
    if dur_beside.get(n):      # same kernel in the product configuration: launches of the backward chain run BESIDE the h-gate data gradient
        d["avg_us_two_stream_backward"] = sum(dur_beside[n]) / len(dur_beside[n])
    if dur.get(n):
        d["avg_us"] = sum(dur[n]) / len(dur[n])
        d["total_ms"] = sum(dur[n]) * 1e-3
    fe, wr = ctr[n].get("FETCH_SIZE"), ctr[n].get("WRITE_SIZE")
    if fe:
        d["fetch_bytes_per_launch"] = 2.0 * 1024.0 * sum(fe) / len(fe)
    if wr:
        d["write_bytes_per_launch"] = 1024.0 * sum(wr) / len(wr)
    if fe and wr and dur.get(n):
        tot = d["fetch_bytes_per_launch"] + d["write_bytes_per_launch"]
        d["hbm_side_GBps"] = tot / (d["avg_us"] * 1e-6) / 1e9
        d["frac_of_8TBps"] = d["hbm_side_GBps"] / 8000.0
    res[n] = d
json.dump({"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) and --kernel-trace over: python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "
                      "(counter passes and avg_us with config async_dgrad off: every kernel alone on the chip; avg_us_two_stream_backward from a trace of the "
                      "product configuration, where the backward chain's launches run beside the h-gate conv's data gradient)",
           "corrections": "FETCH_SIZE / WRITE_SIZE in KB; FETCH_SIZE doubled (gfx950: 128-B requests of 16 B/lane streams tallied at 64 B); "
                          "Infinity-Cache hits included (fabric-side traffic); averages over all launches of a kernel name (all shapes)",
           "kernels": res}, open(out, "w"), indent=1)
for n, d in res.items():
    print(n, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in d.items()})
