#!/bin/bash
# Round 6: launches per step by kernel name (after the fused rank-1 gradients), gap analysis
O=gpurun_out/r06s; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-dense-leg --no-dropin-leg --no-length-leg > $O/bench_prof.json 2> $O/bench_prof.err
f=$(find $O/prof -name "p_kernel_stats.csv" | head -1)
python3 tools/launch_census.py "$f" 6 > $O/launch_census.log 2>&1
t=$(find $O/prof -name "p_kernel_trace.csv" | head -1)
python3 tools/trace_gaps.py "$t" > $O/gaps.log 2>&1
python3 tools/trace_bwd_steps.py "$t" > $O/bwd_steps.log 2>&1
cp "$f" $O/kernel_stats.csv
find $O -name "*trace.csv" -delete
head -50 $O/launch_census.log; head -5 $O/gaps.log
