#!/bin/bash
# Round 6: the fused rank-1 gradient kernel (csrc/rank1_grads.hip) -- unit test, the model tests on its path, same-box A/B against the two GEMMs
O=gpurun_out/r06r; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout 600 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -s -k "rank1_grads_kernel" > $O/pytest_unit.log 2>&1
tail -n 8 $O/pytest_unit.log
export SP_ALLOW_ENV_TUNING=1
for r in 1 2 3; do
  for v in 1 0; do
    SP_RANK1_FUSED=$v timeout 600 python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-dense-leg --no-dropin-leg --no-length-leg 2> $O/err_$v.log | \
      python3 -c "import json,sys; d=[json.loads(l) for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]; print('SP_RANK1_FUSED=$v round $r', d['value'], 'img/s', d['ms_per_step'], 'ms')"
  done
done | tee $O/ab_rank1_fused.log
timeout 1500 python3 -m pytest tests/test_model_gpu.py tests/test_modules_gpu.py -x -q -m gpu -k "bench_path or sparsity_of_the_backward or convlstm or train_step_matches" > $O/pytest_sel.log 2>&1
tail -n 5 $O/pytest_sel.log
