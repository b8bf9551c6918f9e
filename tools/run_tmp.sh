cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04p; mkdir -p $O
timeout 150 python3 tools/tmp_dbg.py 2>&1 | tail -3 | cut -c1-300
timeout 500 python -m pytest tests/test_optim_gpu.py tests/test_ddp_gpu.py -x -q -m gpu > $O/t_optim.log 2>&1; tail -5 $O/t_optim.log
timeout 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "list_attention or deferred_weight or conv2d_fwd_bwd" > $O/t_ops.log 2>&1; tail -3 $O/t_ops.log
timeout 300 python -m pytest tests/test_model_gpu.py -q -m gpu -k "masked_step or train_step_matches_reference" > $O/t_model.log 2>&1; tail -5 $O/t_model.log
for i in 1 2; do timeout 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-dense-leg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; done
timeout 200 python3 tools/bench_backbone.py > $O/backbone.json 2>/dev/null; cut -c1-400 $O/backbone.json
