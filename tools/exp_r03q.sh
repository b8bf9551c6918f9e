#!/bin/bash
O=gpurun_out/r03q; mkdir -p $O
python3 tools/bench_pointwise.py 2>&1 | grep "^{" | cut -c1-120 | tee $O/pointwise_new.log
python3 tools/bench_hconv_quick.py 2>&1 | tail -12 | tee $O/hconv_quick.log
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -5
