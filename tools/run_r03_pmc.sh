#!/bin/bash
# Round-3 counter evidence (on the GPU box, from the repo root).  Separate PMC passes; never combined with trace domains.
O=gpurun_out/r03pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RX='bn_apply_split_kernel|bn_bwd_apply_split_kernel|lstm_bwd_kernel|sum_n_kernel|sum_n_mixed_kernel|clip_adam_kernel'
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$RX" --output-format csv -d $O/hbm/fetch -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/hbm_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$RX" --output-format csv -d $O/hbm/write -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/hbm_write.log 2>&1
rocprofv3 --kernel-trace --kernel-include-regex "$RX" --output-format csv -d $O/hbm/trace -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/hbm_trace.log 2>&1
python3 tools/parse_pmc_hbm.py $O/hbm $O/r03_pmc_hbm_kernels.json > $O/parse_hbm.log 2>&1
i=0
for ctr in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-include-regex 'h2_kernel|hw_kernel' --output-format csv -d $O/pmc_$i -o p -- python3 tools/bench_hconv_fused.py > $O/pmc_$i.log 2>&1
  f=$(find $O/pmc_$i -name "p_counter_collection.csv" | head -1); [ -n "$f" ] && [ "$f" != "$O/pmc_$i/p_counter_collection.csv" ] && cp "$f" $O/pmc_$i/p_counter_collection.csv
done
python3 tools/parse_pmc.py $O/pmc_ 6 $O/r03_pmc_hconv.json > $O/parse_pmc.log 2>&1
python3 tools/bench_hconv_fused.py > $O/hconv_fused.json 2> $O/hconv_fused.err
find $O -name "*.csv" -size +3M -delete
cat $O/parse_hbm.log $O/parse_pmc.log $O/hconv_fused.json
