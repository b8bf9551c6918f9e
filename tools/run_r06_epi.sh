#!/bin/bash
# Round 6: the plain epilogue of h2_kernel without per-element control flow: op tests, pointwise shapes, encoder, bench step
O=gpurun_out/r06ep; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu > $O/pytest_ops.log 2>&1; tail -n 3 $O/pytest_ops.log
python3 tools/bench_pointwise.py 2>/dev/null | grep "^{'shape" | cut -c1-120 | tee $O/pointwise.log
python3 tools/bench_backbone.py 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('backbone fwd', round(d['forward']['ms'],2), 'fwd+bwd', round(d['forward+backward']['ms'],2))" | tee -a $O/pointwise.log
for r in 1 2; do python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-dense-leg --no-dropin-leg --no-length-leg 2>/dev/null | python3 -c "import json,sys; d=[json.loads(l) for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]; print('bench', d['value'], 'img/s', d['ms_per_step'], 'ms')"; done | tee -a $O/pointwise.log
