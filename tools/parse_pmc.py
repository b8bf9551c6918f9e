"""Merge separate rocprofv3 --pmc passes over tools/bench_hconv.py into profiles/<tag>_pmc_hconv.json.
   python tools/parse_pmc.py gpurun_out/pmc_h2_ 4 profiles/r01_pmc_hconv.json
Per kernel (GEMM kernels only, full-size launches of the h-gate conv): counter averages per launch, with the gfx950
corrections of /opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE / WRITE_SIZE in KB; FETCH_SIZE doubled: 128-byte requests of
16 B/lane streams are counted as 64 B; Infinity-Cache hits are included, i.e. this is fabric-side traffic)."""
import collections, csv, json, sys

prefix, npass, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
KEEP = ("h2_kernel", "hw_kernel", "hw2_kernel", "b3_kernel", "w3_kernel", "igemm_kernel", "wgrad_kernel")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for i in range(1, npass + 1):
    for r in csv.DictReader(open(f"{prefix}{i}/p_counter_collection.csv")):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if not name.startswith(KEEP) or int(r["Grid_Size"]) < 100000:
            continue
        name = {"h2_kernel<0>": "h2_kernel<0, 0>", "h2_kernel<1>": "h2_kernel<1, 0>"}.get(name, name)   # passes taken before the DBG template argument
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
kern = {}
for name, cs in acc.items():
    # keep the launches of the big GEMM only (largest grid of each kernel dominates the value counts)
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_side_bytes_per_launch"] = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:      # rocprofv3's MfmaUtil
        d["mfma_busy_pct"] = 100.0 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d:
        d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
    d["launches_averaged"] = {c: len(v) for c, v in cs.items()}
    if name.startswith("hw2_kernel"):
        d["note"] = ("ONE launch = the weight gradient of ALL applications of the h-gate conv of a step (tools/bench_hconv_steps.py: T - 1 = 15 "
                     "segments of M = 81920 pixels): divide bytes / cycles by the segment count to compare with a per-application launch")
    kern[name] = d
prev = {}
try:
    prev = json.load(open(out)).get("kernels", {})
except Exception:
    pass
prev.update(kern)
json.dump({"command": "rocprofv3 --pmc <counter set> --output-format csv -- python3 tools/bench_hconv.py  (separate passes: FETCH_SIZE | "
                      "WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum | GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES | "
                      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE); merged by tools/parse_pmc.py",
           "workload": "h-gate conv3x3 512->2048 at B=32, 40x64 (M=81920, N=2048, K=4608): fwd, dgrad, wgrad; averages per launch",
           "corrections": "FETCH_SIZE, WRITE_SIZE in KB; FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B for 16 B/lane streams); "
                          "Infinity-Cache hits are included in FETCH_SIZE (guide section HBM) -- fabric-side, not pure HBM, traffic",
           "kernels": prev}, open(out, "w"), indent=1)
for k, d in kern.items():
    print(k, {c: (round(v, 1) if isinstance(v, float) else v) for c, v in d.items() if c != "launches_averaged"})
