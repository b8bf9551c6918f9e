#!/bin/bash
O=gpurun_out/r03h
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SP_LIBRARY=timing N_ITER=6
run() {   # tag, env assignments exported by the caller
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-include-regex 'h2_kernel' --output-format csv -d $O/$1/p1 -o p -- python3 tools/bench_hconv_quick.py > $O/$1.p1.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex 'h2_kernel' --output-format csv -d $O/$1/p2 -o p -- python3 tools/bench_hconv_quick.py > $O/$1.p2.log 2>&1
  python3 tools/pmc_simple.py $O/$1 > $O/$1.txt 2>&1
}
export SP_H2_HALO=1; unset SP_H2_DBG; run halo
export SP_H2_HALO=0; run nohalo
export SP_H2_HALO=0 SP_H2_DBG=11; run proxy
unset SP_H2_DBG
find $O -name "*.csv" -delete
for t in halo nohalo proxy; do echo "== $t"; cat $O/$t.txt; done
