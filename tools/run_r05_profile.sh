#!/bin/bash
# Round 5: kernel trace of the bench line (7 steps) -> per-shape table, kernel stats and the gap analysis of tools/trace_gaps.py
O=gpurun_out/r05p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dense-leg --no-dropin-leg > $O/bench_prof.json 2> $O/bench_prof.err
f=$(find $O/prof_bench -name "p_kernel_trace.csv" | head -1)
python3 tools/summarize_prof.py "${f%_kernel_trace.csv}" $O/r05 > $O/summ_bench.log 2>&1
python3 tools/trace_gaps.py "$f" > $O/gaps.log 2>&1
find $O -name "*trace.csv" -delete
cat $O/gaps.log
