cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04v; mkdir -p $O
timeout 200 python3 tools/encoder_census.py --only wgrad > $O/census_new.log 2>&1; head -2 $O/census_new.log | tail -1; grep "^wgrad" $O/census_new.log | cut -c1-120
timeout 400 python -m pytest tests/test_ops_gpu.py -q -m gpu > $O/t_ops.log 2>&1; tail -3 $O/t_ops.log
timeout 300 python -m pytest tests/test_model_gpu.py -q -m gpu -k "masked_step or train_step_matches_reference" > $O/t_model.log 2>&1; tail -3 $O/t_model.log
timeout 200 python3 tools/bench_backbone.py > $O/backbone.json 2>/dev/null; python3 -c "import json; d=json.load(open('$O/backbone.json')); print(d['forward']['ms'], d['forward+backward']['ms'])"
for i in 1 2; do timeout 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-dense-leg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; done
