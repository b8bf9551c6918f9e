#!/bin/bash
# Round 6: same-box A/B of the branch-free GEMM epilogue (this tree) against the commit before it (_prev_tree/), then the model tests on its path
O=gpurun_out/r06eq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/ab_prev_tree.sh "" 3 > $O/ab_epilogue.log 2>&1; cat $O/ab_epilogue.log
(cd _prev_tree && python3 tools/bench_backbone.py 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('prev backbone', round(d['forward']['ms'],2), round(d['forward+backward']['ms'],2))") | tee -a $O/ab_epilogue.log
python3 tools/bench_backbone.py 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('this backbone', round(d['forward']['ms'],2), round(d['forward+backward']['ms'],2))" | tee -a $O/ab_epilogue.log
timeout 1500 python3 -m pytest tests/test_model_gpu.py tests/test_modules_gpu.py -x -q -m gpu -k "bench_path or sparsity_of_the_backward or train_step_matches or tame_all_steps or encoder_fusions or eval_forward" > $O/pytest_sel.log 2>&1; tail -n 3 $O/pytest_sel.log
