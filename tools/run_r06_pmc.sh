#!/bin/bash
# Round-6 counter evidence (same dominant kernels as round 5; taken on the head build so that roofline.traffic describes the shipped code) (on the GPU box, from the repo root): six separate PMC passes over the dominant launches as the step issues them
# (fused forward, data gradient, the ONE deferred weight-gradient launch hw2_kernel).  Never combined with trace domains.
O=gpurun_out/r06pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for ctr in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  N_ITER=2 rocprofv3 --pmc $ctr --kernel-include-regex 'h2_kernel|hw_kernel|hw2_kernel' --output-format csv -d $O/pmc_$i -o p -- python3 tools/bench_hconv_steps.py > $O/pmc_$i.log 2>&1
  f=$(find $O/pmc_$i -name "p_counter_collection.csv" | head -1); [ -n "$f" ] && [ "$f" != "$O/pmc_$i/p_counter_collection.csv" ] && cp "$f" $O/pmc_$i/p_counter_collection.csv
done
python3 tools/parse_pmc.py $O/pmc_ 6 $O/r06_pmc_hconv.json > $O/parse_pmc.log 2>&1
python3 tools/bench_hconv_steps.py > $O/hconv_steps.json 2> $O/hconv_steps.err
find $O -name "*.csv" -size +3M -delete
cat $O/parse_pmc.log $O/hconv_steps.json
