#!/bin/bash
# Round 6: same-box A/B of a kernel change (this tree) against the commit before it (_prev_tree/): encoder, step; op + model tests on its path
O=gpurun_out/r06eq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu > $O/pytest_ops.log 2>&1; tail -n 2 $O/pytest_ops.log
for r in 1 2 3; do
(cd _prev_tree && python3 tools/bench_backbone.py 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('prev backbone', round(d['forward']['ms'],2), round(d['forward+backward']['ms'],2))")
python3 tools/bench_backbone.py 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('this backbone', round(d['forward']['ms'],2), round(d['forward+backward']['ms'],2))"
done | tee $O/ab_kernel.log
python3 tools/encoder_census.py 2>/dev/null | grep "wgrad" | head -12 | cut -c1-150 | tee -a $O/ab_kernel.log
timeout 1500 python3 -m pytest tests/test_model_gpu.py -x -q -m gpu -k "bench_path or train_step_matches or encoder_fusions" > $O/pytest_sel.log 2>&1; tail -n 2 $O/pytest_sel.log
