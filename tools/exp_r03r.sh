#!/bin/bash
O=gpurun_out/r03r; mkdir -p $O
SP_LIBRARY=timing python3 tools/bench_drt.py 2>&1 | grep "^{" | tee $O/drt_old.json
python3 tools/bench_drt.py 2>&1 | grep "^{" | tee $O/drt_new.json
timeout 600 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "direct_head" 2>&1 | tail -3
