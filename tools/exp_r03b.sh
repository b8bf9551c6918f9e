#!/bin/bash
export SP_LIBRARY=timing      # knobs below exist in libscanpaths_amd_timing.so only (make -C scanpaths_amd/csrc timing)
# round-3 experiment batch: gradient error table, accumulation-level accuracy, weight-side LDS traffic proxy
O=gpurun_out/r03b
mkdir -p $O
python tests/diagnostics/grad_error_table.py 4 > $O/grad_table_T4.log 2>&1
for dbg in 12 10 6 8; do
  SP_H2_DBG=$dbg python tools/bench_hconv_quick.py > $O/hconv_dbg$dbg.json 2> $O/hconv_dbg$dbg.err
done
SP_H2_CHUNK=1048576 python tools/bench_hconv_quick.py > $O/hconv_chunkinf.json 2>&1
SP_H2_CHUNK=1048576 python -m pytest tests/test_ops_gpu.py -m gpu -q -s -k "hgate_conv_at_benchmark_size and f16x2 or split_gemms_are_as_accurate" > $O/chunkinf_ops.log 2>&1
python -m pytest tests/test_ops_gpu.py -m gpu -q -s -k "hgate_conv_at_benchmark_size and f16x2 or split_gemms_are_as_accurate" > $O/chunk8_ops.log 2>&1
SP_H2_CHUNK=1048576 python -m pytest tests/test_model_gpu.py -m gpu -q -s -k "tame_all_steps and f16x2" > $O/chunkinf_tame.log 2>&1
tail -3 $O/*.log; cat $O/hconv_*.json
