#!/bin/bash
# Round 6 probe: the short-K pointwise GEMMs under the timing build's modes (wrong results): 6 no loads, 7 no MFMAs, 8 MFMAs only, 9 LDS-DMA loads only;
# and (first version of this script, profiles/r06_pointwise_stagger.log) a first-round stagger of the CUs: no effect
O=gpurun_out/r06y; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export SP_ALLOW_ENV_TUNING=1 SP_LIBRARY=timing
for v in 0 6 7 8 9; do
  echo "== SP_H2_DBG=$v"
  SP_H2_DBG=$v python3 tools/bench_pointwise.py 2>/dev/null | grep "^{'shape" | cut -c1-120
done | tee $O/modes.log
