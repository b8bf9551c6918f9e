#!/bin/bash
# Round 5: how does FETCH_SIZE tally the fused-cell epilogue's loads (16 B per lane, 64-byte runs per pixel and gate)?  The guide's gfx950
# correction (double FETCH_SIZE) holds for wide coalesced streams whose 128-byte requests are tallied at 64 B; other widths are
# uncalibrated.  Known byte count: the epilogue reads 5120 tiles x (128 KB x-gates + 32 KB c_prev) = 0.839 GB per launch.  Two PMC passes
# over the same launch of the timing library, with (probe 0) and without (probe 1) the epilogue's loads: the difference of the raw counts
# is what 0.839 GB of such loads add.
O=gpurun_out/r05calib; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SP_ALLOW_ENV_TUNING=1 SP_LIBRARY=timing N_ITER=4
for v in 0 1; do
  export SP_H2_DBG=$v
  rocprofv3 --pmc FETCH_SIZE --kernel-include-regex 'h2_kernel' --output-format csv -d $O/p$v -o p -- python3 tools/bench_hconv_fused.py > $O/p$v.log 2>&1
  f=$(find $O/p$v -name "p_counter_collection.csv" | head -1)
  python3 - "$f" $v <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "true, true, true" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
v = [float(r["Counter_Value"]) for r in rows]
print(f"probe {sys.argv[2]}: fused forward launches {len(v)}, FETCH_SIZE raw mean {sum(v) / len(v) / 1e6:.4f} GB (KB units / 1e6)")
PY
done | tee $O/calib.log
find $O -name "*.csv" -size +3M -delete
