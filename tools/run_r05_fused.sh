#!/bin/bash
# Round 5: where the fused-cell forward's epilogue time goes (timing library, sp_set_tuning("h2_dbg", n) probes of h2_kernel<fwd, LSTM>):
#   1 = no epilogue loads, 2 = no epilogue stores, 3 = neither, 4 = no epilogue, 8 = streaming (nt) stores, 16 = streaming loads;
#   (G << 8) | (step << 12): first-round stagger, G groups of CUs start (group) x step x 2 us late.  Same box, interleaved.
# Usage: gpurun -- bash tools/run_r05_fused.sh "0 8 16 24"
O=gpurun_out/r05c; mkdir -p $O
export SP_ALLOW_ENV_TUNING=1 SP_LIBRARY=timing N_ITER=20
for r in 1 2 3; do
  for v in ${1:-0 1 2 3 4}; do
    SP_H2_DBG=$v python3 tools/bench_hconv_fused.py 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('probe $v round $r fused fwd', d['h2_fwd']['avg_ms'], 'ms  dgrad', d['h2_dgrad']['avg_ms'])"
  done
done | tee -a $O/probe.log
