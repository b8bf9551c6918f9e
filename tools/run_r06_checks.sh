#!/bin/bash
# Round 6: the new parity legs (per-module goldens, full-batch bs 32, side-stream without accidental keepers), the CPU-baseline thread
# sweep (one-off, committed as profiles/r06_cpu_thread_sweep.json) and the bench line with its new fields
O=gpurun_out/r06b; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
( python3 bench.py --cpu-sweep > $O/cpu_sweep.json 2> $O/cpu_sweep.err ) &
SWEEP=$!
python3 -m pytest tests/test_modules_gpu.py -x -q -m gpu > $O/pytest_modules.log 2>&1
python3 -m pytest tests/test_model_gpu.py -x -q -m gpu -k "side_stream or full_batch or sparsity_of_the_backward" -s > $O/pytest_model_sel.log 2>&1
python3 -m pytest tests/test_ops_gpu.py tests/test_ddp_gpu.py tests/test_optim_gpu.py -x -q -m gpu > $O/pytest_ops.log 2>&1
wait $SWEEP
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --force-bucketer --no-dense-leg > $O/bench_bucketer.json 2> $O/bench_bucketer.err
tail -3 $O/pytest_modules.log $O/pytest_model_sel.log $O/pytest_ops.log; cat $O/cpu_sweep.json; tail -2 $O/cpu_sweep.err
