#!/bin/bash
O=gpurun_out/r03i
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SP_LIBRARY=timing N_ITER=6 SP_H2_HALO=0
run() {
  rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --kernel-include-regex 'h2_kernel' --output-format csv -d $O/$1/p1 -o p -- python3 tools/bench_hconv_quick.py > $O/$1.p1.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex 'h2_kernel' --output-format csv -d $O/$1/p2 -o p -- python3 tools/bench_hconv_quick.py > $O/$1.p2.log 2>&1
  rocprofv3 --kernel-trace --kernel-include-regex 'h2_kernel' --output-format csv -d $O/$1/p3 -o p -- python3 tools/bench_hconv_quick.py > $O/$1.p3.log 2>&1
  python3 tools/pmc_simple.py $O/$1 > $O/$1.txt 2>&1
  python3 - "$O/$1" >> $O/$1.txt <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p3/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "h2_kernel" in r["Kernel_Name"] and int(r.get("Grid_Size", "0") or 0) >= 500000:
            d[r["Kernel_Name"].split("(")[0][-40:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
for k, v in d.items():
    print("TRACE", k, "avg ms", round(sum(v) / len(v), 4), "n", len(v))
PY
}
unset SP_H2_DBG; run full
export SP_H2_DBG=8; run mfma_only
export SP_H2_DBG=6; run no_loads
export SP_H2_DBG=7; run no_mfma
unset SP_H2_DBG
find $O -name "*.csv" -delete
for t in full mfma_only no_loads no_mfma; do echo "== $t"; cat $O/$t.txt | cut -c1-700; done
