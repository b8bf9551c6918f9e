#!/bin/bash
# Round 6: the batched duration branch -- parity (model goldens, module goldens, equivalence with the per-step form) and same-box A/B
O=gpurun_out/r06d; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
python3 -m pytest tests/test_modules_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "head_conv or batched_duration or sparsity_of_the_backward or second_consumer or tame_all_steps or train_step_matches or eval_forward" > $O/pytest_sel.log 2>&1
tail -n 5 $O/pytest_sel.log
export SP_ALLOW_ENV_TUNING=1
for r in 1 2 3; do
  for v in 1 0; do
    SP_DRT_BATCHED=$v python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-dense-leg --no-dropin-leg --no-length-leg 2> $O/err_$v.log | \
      python3 -c "import json,sys; d=[json.loads(l) for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]; print('SP_DRT_BATCHED=$v round $r', d['value'], 'img/s', d['ms_per_step'], 'ms')"
  done
done | tee $O/ab_drt_batched.log
