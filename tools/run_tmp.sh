cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04n; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -q -m gpu -k "fan_in or direct_head or semantic_pool or live_first" > $O/t_ops.log 2>&1; tail -3 $O/t_ops.log
python -m pytest tests/test_model_gpu.py -q -m gpu -k "masked_step or train_step_matches_reference" > $O/t_model.log 2>&1; tail -5 $O/t_model.log
for i in 1 2; do python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-dense-leg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; done
