#!/bin/bash
# Round 6: is the device ever waiting for the host?  empty intervals of a traced step; host-side profile of one step
O=gpurun_out/r06w; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d $O/prof -o p -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-dense-leg --no-dropin-leg --no-length-leg > $O/bench_prof.json 2> $O/bench_prof.err
t=$(find $O/prof -name "p_kernel_trace.csv" | head -1)
python3 tools/trace_idle.py "$t" > $O/idle.log 2>&1
find $O -name "*trace.csv" -delete
cat $O/idle.log
python3 tools/host_profile.py > $O/host_profile.log 2>&1; head -70 $O/host_profile.log
