#!/bin/bash
O=gpurun_out/r03j
mkdir -p $O
python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "direct_head" > $O/ops.log 2>&1
python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "bench_path or train_step_matches_reference" > $O/model.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --kernel-include-regex 'drt_|lstm_bwd' --output-format csv -d $O/prof -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
f=$(find $O/prof -name "p_kernel_stats.csv" | head -1); head -8 "$f" | cut -d, -f1-6
find $O -name "*trace.csv" -delete
tail -n 3 $O/ops.log $O/model.log
