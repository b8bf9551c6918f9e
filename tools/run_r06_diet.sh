#!/bin/bash
# Round 6: the per-step PyTorch copies removed (pooled rows [S,B,C], strided logits gradient, one-launch wc^T split): tests, launches per step, A/B n/a
O=gpurun_out/r06t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "semantic_pool or launch_savers or rank1_grads or direct_head or list_attention" > $O/pytest_ops.log 2>&1; tail -n 3 $O/pytest_ops.log
timeout 1500 python3 -m pytest tests/test_model_gpu.py tests/test_modules_gpu.py -x -q -m gpu -k "bench_path or sparsity_of_the_backward or train_step_matches or head_conv or batched_duration or tame_all_steps" > $O/pytest_sel.log 2>&1; tail -n 3 $O/pytest_sel.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-dense-leg --no-dropin-leg --no-length-leg > $O/bench_prof.json 2> $O/bench_prof.err
f=$(find $O/prof -name "p_kernel_stats.csv" | head -1)
python3 tools/launch_census.py "$f" 6 > $O/launch_census.log 2>&1
t=$(find $O/prof -name "p_kernel_trace.csv" | head -1)
python3 tools/trace_gaps.py "$t" > $O/gaps.log 2>&1
find $O -name "*trace.csv" -delete
head -4 $O/launch_census.log; head -3 $O/gaps.log
for r in 1 2; do python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-dense-leg --no-dropin-leg --no-length-leg 2>/dev/null | python3 -c "import json,sys; d=[json.loads(l) for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]; print('bench', d['value'], 'img/s', d['ms_per_step'], 'ms')"; done
