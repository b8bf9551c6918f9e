#!/bin/bash
# six PMC passes over tools/bench_hconv.py with the current default kernels -> gpurun_out/$1/r02_pmc_hconv.json
O=gpurun_out/${1:-r02n}; mkdir -p $O; cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for ctr in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_$i -o p -- python3 tools/bench_hconv.py > $O/pmc_$i.log 2>&1
  f=$(find $O/pmc_$i -name "p_counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/pmc_$i/p_counter_collection.csv
done
python3 tools/parse_pmc.py $O/pmc_ 6 $O/r02_pmc_hconv.json | grep -E "h2_kernel|hw_kernel" | cut -c1-420
