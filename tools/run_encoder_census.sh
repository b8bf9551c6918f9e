#!/bin/bash
# GPU box: per-GEMM census of the encoder + kernel trace of the encoder-only bench
O=gpurun_out/r04enc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 tools/encoder_census.py > $O/census.log 2> $O/census.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 tools/bench_backbone.py --reps 4 > $O/backbone.json 2> $O/backbone.err
f=$(find $O/prof -name "p_kernel_trace.csv" | head -1); python3 tools/summarize_prof.py "${f%_kernel_trace.csv}" $O/enc > $O/summ.log 2>&1
find $O -name "*trace.csv" -delete
head -50 $O/census.log
