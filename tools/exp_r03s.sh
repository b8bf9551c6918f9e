#!/bin/bash
O=gpurun_out/r03s; mkdir -p $O
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 1500 python3 -m pytest tests/test_model_gpu.py -x -q -m gpu -k "bench_path or train_step_matches" 2>&1 | tail -3
for r in 1 2; do
  for v in 1 0; do
    SP_LSTM_SKIP_DPRE=$v python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2> $O/bench_err.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skip_dpre=$v round $r', d['value'], d['ms_per_step'])"
  done
done
tail -3 $O/bench_err.log
