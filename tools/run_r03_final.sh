#!/bin/bash
# Round-3 artifact run (GPU box, repo root): full -m gpu suite, bench lines, kernel trace, PMC passes, CPU-baseline protocol run.
O=gpurun_out/r03final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_nocpu.json 2> $O/bench_nocpu.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
f=$(find $O/prof_bench -name "p_kernel_trace.csv" | head -1); python3 tools/summarize_prof.py "${f%_kernel_trace.csv}" $O/r03 > $O/summ_bench.log 2>&1
find $O -name "*trace.csv" -delete
python3 tools/bench_backbone.py > $O/backbone.json 2> $O/backbone.err
python3 examples/train_synthetic.py > $O/example.log 2>&1; echo "example rc=$?" >> $O/example.log
# (run once this round: profiles/r03_cpu_baseline.json)  python3 bench.py --steps 3 --warmup 1 --cpu-batch 4 --cpu-threads 128 --cpu-steps 2 > $O/cpu_baseline_b4_t128.json 2> $O/cpu_b4_t128.err
# (run once this round: profiles/r03_cpu_baseline.json)  python3 bench.py --steps 3 --warmup 1 --cpu-batch 4 --cpu-threads 32 --cpu-steps 2 > $O/cpu_baseline_b4_t32.json 2> $O/cpu_b4_t32.err
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -n 4 $O/pytest.log; tail -n 3 $O/example.log; cut -c1-300 $O/backbone.json; cut -c1-260 $O/bench.json
