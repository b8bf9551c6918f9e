#!/bin/bash
O=gpurun_out/r03m
mkdir -p $O
export SP_LIBRARY=timing
for tb in 0 1 0 1; do
  SP_TWOBAR=$tb python tools/bench_hconv_quick.py >> $O/quick_tb$tb.json 2>> $O/quick.err
  SP_TWOBAR=$tb python tools/bench_hconv_fused.py >> $O/fused_tb$tb.json 2>> $O/fused.err
done
SP_TWOBAR=1 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "conv2d_fwd_bwd and f16x2 or lstm_cell_and_gate_conv or hgate_conv_at_benchmark_size and f16x2" > $O/ops_tb1.log 2>&1
for t in 0 1; do echo "== twobar $t"; cat $O/quick_tb$t.json $O/fused_tb$t.json | cut -c1-330; done; tail -n 3 $O/ops_tb1.log
