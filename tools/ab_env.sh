#!/bin/bash
# Same-box interleaved A/B of one environment switch on the bench line (box-to-box spread is +-1.5 %, larger than most levers):
#   bash tools/ab_env.sh SP_LSTM_SKIP_DPRE "1 0" [rounds=3] [extra bench.py args]
# e.g. SP_RANK1_DSP_SPLIT, SP_RANK1_DWC_SPLIT, SP_LSTM_SKIP_DPRE, SP_BN_SKIP_DX, SP_DEFER_WGRAD, SP_CHANNEL_SCALES, or with SP_LIBRARY=timing exported: SP_H2_HALO, SP_HW_SPLITS.
# (scanpaths_amd.config honours SP_* switches only under SP_ALLOW_ENV_TUNING=1; the JSON line lists them as non_default_switches)
export SP_ALLOW_ENV_TUNING=1
VAR=$1; VALUES=$2; ROUNDS=${3:-3}; shift 3 2>/dev/null
O=gpurun_out/ab_env; mkdir -p $O
for r in $(seq 1 $ROUNDS); do
  for v in $VALUES; do
    env $VAR=$v python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2> $O/err.log | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v round $r', d['value'], 'img/s', d['ms_per_step'], 'ms')"
  done
done | tee $O/${VAR}.log
