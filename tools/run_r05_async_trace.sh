#!/bin/bash
# Round 5: raw kernel time line of two backward decode steps around the h-gate conv's data gradient (side stream, queue 2)
O=gpurun_out/r05q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d $O/prof -o p -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-dense-leg --no-dropin-leg > $O/bench.json 2> $O/bench.err
f=$(find $O/prof -name "p_kernel_trace.csv" | head -1)
for n in ${@:-44}; do python3 tools/trace_window.py "$f" "h2_kernel<1, 3, true, false, true" $n > $O/window_bwd_$n.log 2>&1; done
cp $O/window_bwd_${1:-44}.log $O/window_bwd.log
python3 tools/trace_gaps.py "$f" > $O/gaps.log 2>&1
find $O -name "*trace.csv" -delete
head -120 $O/window_bwd.log
