#!/bin/bash
# Round 6 probe: the short-K pointwise GEMMs under the timing build's modes (wrong results): 6 no loads, 7 no MFMAs, 8 MFMAs only, 9 LDS-DMA loads only,
# 10 no global stores in the epilogue, 11 no epilogue; (first version of this script, profiles/r06_pointwise_stagger.log: a first-round stagger of the CUs)
O=gpurun_out/r06y; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export SP_ALLOW_ENV_TUNING=1 SP_LIBRARY=timing
for v in 0 11 10; do
  echo "== SP_H2_DBG=$v"
  SP_H2_DBG=$v python3 tools/bench_pointwise.py 2>/dev/null | grep "^{'shape" | cut -c1-120
done | tee $O/modes2.log
