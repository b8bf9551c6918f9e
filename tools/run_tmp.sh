cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dense-leg > $O/bench_prof.json 2> $O/bench_prof.err
f=$(find $O/prof_bench -name "p_kernel_trace.csv" | head -1); python3 tools/summarize_prof.py "${f%_kernel_trace.csv}" $O/r04s > $O/summ_bench.log 2>&1
find $O -name "*trace.csv" -delete
