#!/bin/bash
O=gpurun_out/r03g
mkdir -p $O
python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "conv2d_fwd_bwd and f16x2 or lstm_cell_and_gate_conv or hgate_conv_at_benchmark_size and f16x2 or fused_gateconv" > $O/ops.log 2>&1
SP_LIBRARY=timing python tools/bench_hconv_quick.py > $O/hconv_halo.json 2> $O/hconv_halo.err
SP_LIBRARY=timing SP_H2_HALO=0 python tools/bench_hconv_quick.py > $O/hconv_nohalo.json 2> $O/hconv_nohalo.err
SP_LIBRARY=timing python tools/bench_hconv_fused.py > $O/fused_halo.json 2> $O/fused_halo.err
SP_LIBRARY=timing SP_H2_HALO=0 python tools/bench_hconv_fused.py > $O/fused_nohalo.json 2> $O/fused_nohalo.err
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -n 15 $O/ops.log; cat $O/hconv_halo.json $O/hconv_nohalo.json $O/fused_halo.json $O/fused_nohalo.json; cut -c1-250 $O/bench.json
