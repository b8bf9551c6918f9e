#!/bin/bash
O=gpurun_out/r03n
mkdir -p $O
python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "direct_head" > $O/ops.log 2>&1
python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "train_step_matches_reference or tame_all_steps and f16x2" > $O/model.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --kernel-include-regex 'drt_' --output-format csv -d $O/prof -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
f=$(find $O/prof -name "p_kernel_stats.csv" | head -1); cut -d, -f1-4 "$f" | cut -c1-120
find $O -name "*trace.csv" -delete
tail -n 3 $O/ops.log $O/model.log; cut -c1-200 $O/bench_prof.json
