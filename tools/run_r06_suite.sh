#!/bin/bash
# Round 6: the whole GPU suite (durations), then a kernel trace of the bench step: launches per step, gaps between fused launches, which
# side of the two-stream backward is critical
O=gpurun_out/r06e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 -m pytest tests -q -m gpu --durations=20 > $O/pytest_gpu.log 2>&1
tail -n 30 $O/pytest_gpu.log
rocprofv3 --kernel-trace --output-format csv -d $O/prof -o p -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-dense-leg --no-dropin-leg --no-length-leg > $O/bench_prof.json 2> $O/bench_prof.err
f=$(find $O/prof -name "p_kernel_trace.csv" | head -1)
python3 tools/trace_bwd_steps.py "$f" > $O/bwd_steps.log 2>&1
python3 tools/trace_gaps.py "$f" > $O/gaps.log 2>&1
python3 tools/trace_window.py "$f" "h2_kernel<0, 3, true, true, true" 20 > $O/window_fwd_20.log 2>&1
find $O -name "*trace.csv" -delete
head -8 $O/gaps.log; tail -4 $O/bwd_steps.log | head -2
