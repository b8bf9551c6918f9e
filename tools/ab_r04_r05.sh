#!/bin/bash
# Same-box, interleaved: the round-4 tree (git archive 0817bc3 -> _r04_tree/, its own library built there) against this tree.
#   rm -rf _r04_tree && mkdir _r04_tree && git archive 0817bc3 | tar -x -C _r04_tree && make -C _r04_tree/scanpaths_amd/csrc -j8
#   gpurun -- bash tools/ab_r04_r05.sh
O=gpurun_out/r05ab; mkdir -p $O
for r in 1 2 3; do
  (cd _r04_tree && python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('r04 round $r', d['value'], 'img/s', d['ms_per_step'], 'ms; dense', d['backward_sparsity']['dense_backward_ms_per_step'], '; fused fwd', d['roofline']['avg_launch_ms'])")
  python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('r05 round $r', d['value'], 'img/s', d['ms_per_step'], 'ms; dense', d['backward_sparsity']['dense_backward_ms_per_step'], '; drop-in', d['dropin']['dropin_ms_per_step'], '; fused fwd', d['roofline']['avg_launch_ms'])"
done | tee $O/r04_vs_r05.log
