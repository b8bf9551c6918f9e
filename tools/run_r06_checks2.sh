#!/bin/bash
O=gpurun_out/r06c; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
python3 bench.py --cpu-sweep > $O/cpu_sweep.json 2> $O/cpu_sweep.err
python3 -X faulthandler bench.py --steps 5 --warmup 2 --no-cpu-baseline --force-bucketer --no-dense-leg > $O/bench_bucketer.json 2> $O/bench_bucketer.err; echo "bucketer rc=$?" >> $O/bench_bucketer.err
python3 tools/chain_ablation.py --steps 6 --rounds 2 > $O/chain_ablation.json 2> $O/chain_ablation.err
python3 -m pytest tests/test_model_gpu.py -x -q -m gpu -k "accidental_keepers or full_batch" -s > $O/pytest_model_sel.log 2>&1
tail -n 3 $O/pytest_model_sel.log; cat $O/cpu_sweep.json; tail -n 4 $O/bench_bucketer.err; cat $O/chain_ablation.json
