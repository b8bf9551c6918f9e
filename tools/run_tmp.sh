cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04w; mkdir -p $O
timeout 200 python3 tools/encoder_census.py --only wgrad_multi > $O/census_new.log 2>&1; cat $O/census_new.log | cut -c1-125
timeout 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "deferred or hgate" > $O/t_ops.log 2>&1; tail -3 $O/t_ops.log
timeout 300 python -m pytest tests/test_model_gpu.py -q -m gpu -k "masked_step or train_step_matches_reference" > $O/t_model.log 2>&1; tail -3 $O/t_model.log
for i in 1 2; do timeout 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-dense-leg 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])
for g in d['roofline']['timed_gemms']: print('   ', g['kernel'][:34], g['M'], g['N'], g['K'], g['launches_per_step'], g['avg_launch_ms'], g['tflops'])"; done
