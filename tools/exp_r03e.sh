#!/bin/bash
O=gpurun_out/r03e
mkdir -p $O
DETAIL=sal_conv.weight,sal_conv.bias,performance_sal_layer.True.weight,object_head.sal_layer_3.bias CONFIGS=bench_path,bf16x3 NROWS=4 python tests/diagnostics/grad_error_table.py 16 > $O/grad_T16b.log 2>&1
SP_LIBRARY=timing python tools/bench_hconv_quick.py > $O/hconv_default.json 2> $O/hconv_default.err
SP_LIBRARY=timing SP_H2_DBG=10 python tools/bench_hconv_quick.py > $O/hconv_dbg10.json 2> $O/hconv_dbg10.err
SP_LIBRARY=timing SP_H2_DBG=5 python tools/bench_hconv_quick.py > $O/hconv_dbg5.json 2> $O/hconv_dbg5.err
python -m pytest tests/test_dataset_eval_gpu.py tests/test_inference_gpu.py tests/test_ops_gpu.py -m gpu -q -s -k "dataset_variants or validation_metrics_match or config5 or product_library or conv_epilogue_writes" > $O/newtests.log 2>&1
bash tools/run_r03_pmc.sh > $O/pmc.log 2>&1
grep -E "DETAIL|==" $O/grad_T16b.log | cut -c1-700; cat $O/hconv_*.json; tail -n 5 $O/newtests.log; tail -n 30 $O/pmc.log
