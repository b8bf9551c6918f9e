#!/bin/bash
# Round 6: the tile loop of the short-K pointwise GEMMs (h2_kernel PERSIST builds): op tests, A/B on the pointwise shapes and on the encoder
O=gpurun_out/r06z; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv2d or bn_ or epilogue or stats or pointwise" > $O/pytest_ops.log 2>&1; tail -n 3 $O/pytest_ops.log
export SP_ALLOW_ENV_TUNING=1 SP_LIBRARY=timing
for r in 1 2; do for v in 1 0; do
  echo "== SP_H2_PERSIST=$v round $r"
  SP_H2_PERSIST=$v python3 tools/bench_pointwise.py 2>/dev/null | grep "^{'shape" | cut -c1-120
  SP_H2_PERSIST=$v python3 tools/bench_backbone.py 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('backbone fwd', round(d['forward']['ms'],2), 'fwd+bwd', round(d['forward+backward']['ms'],2))"
done; done | tee $O/ab_persist.log
