#!/bin/bash
O=gpurun_out/r03u; mkdir -p $O
for r in 1 2; do
  for v in 2048 1024 3072 4096; do
    SP_LIBRARY=timing SP_HW_SPLITS=$v python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2> $O/bench_err.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hw target=$v round $r', d['value'], d['ms_per_step'])"
  done
done
