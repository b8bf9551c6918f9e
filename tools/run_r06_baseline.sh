#!/bin/bash
# Round 6: baseline of the tree on this box -- bench line (no CPU leg), kernel trace of 3 steps, which side of the two-stream backward
# recurrence is the critical path (tools/trace_bwd_steps.py), gap analysis.
O=gpurun_out/r06a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --output-format csv -d $O/prof -o p -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-dense-leg --no-dropin-leg > $O/bench_prof.json 2> $O/bench_prof.err
f=$(find $O/prof -name "p_kernel_trace.csv" | head -1)
python3 tools/trace_bwd_steps.py "$f" > $O/bwd_steps.log 2>&1
python3 tools/trace_gaps.py "$f" > $O/gaps.log 2>&1
python3 tools/trace_window.py "$f" "h2_kernel<1, 3, true, false, true" 40 > $O/window_bwd_40.log 2>&1
python3 tools/trace_window.py "$f" "h2_kernel<0, 3, true, true, true" 20 > $O/window_fwd_20.log 2>&1
find $O -name "*trace.csv" -delete
cat $O/bench.json; cat $O/bwd_steps.log
