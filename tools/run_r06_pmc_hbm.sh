#!/bin/bash
# Round 6: HBM-side counters of the HBM-bound kernels as the code stands now (profiles/r03_pmc_hbm_kernels.json described
# round 3's kernels).  Separate passes, never combined with trace domains: kernel trace (product configuration), kernel trace / FETCH_SIZE /
# WRITE_SIZE with the two-stream backward off (every kernel alone on the chip: stand-alone durations and clean counters).
O=gpurun_out/r06hbm; mkdir -p $O/trace_async $O/trace $O/fetch $O/write
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dense-leg --no-dropin-leg --no-length-leg"
rocprofv3 --kernel-trace --output-format csv -d $O/trace_async -o p -- python3 $ARGS > $O/trace_async.json 2> $O/trace_async.err
export SP_ALLOW_ENV_TUNING=1 SP_ASYNC_DGRAD=0
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o p -- python3 $ARGS > $O/trace.json 2> $O/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python3 $ARGS > $O/fetch.json 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python3 $ARGS > $O/write.json 2> $O/write.err
python3 tools/parse_pmc_hbm.py $O $O/r06_pmc_hbm_kernels.json > $O/parse.log 2>&1
find $O -name "*.csv" -delete
cat $O/parse.log
