cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04k; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -q -m gpu -k "live_first or deferred_weight" > $O/t_ops.log 2>&1; tail -3 $O/t_ops.log
python -m pytest tests/test_model_gpu.py -q -m gpu -k "masked_step" > $O/t_model.log 2>&1; tail -3 $O/t_model.log
SP_LIBRARY=timing bash tools/ab_env.sh SP_ROW_ORDER "1 0" 2 > $O/ab.log 2>&1; cat $O/ab.log
