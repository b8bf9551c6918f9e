#!/bin/bash
O=gpurun_out/r03f
mkdir -p $O
python tests/diagnostics/small_value_precision.py > $O/small_values.log 2>&1
SP_LIBRARY=timing python tools/bench_hconv_quick.py > $O/hconv_default.json 2> $O/hconv_default.err
SP_LIBRARY=timing SP_H2_DBG=11 python tools/bench_hconv_quick.py > $O/hconv_dbg11.json 2> $O/hconv_dbg11.err
SP_LIBRARY=timing python tools/bench_hconv_quick.py > $O/hconv_default2.json 2> $O/hconv_default2.err
python tools/bench_pointwise.py > $O/pointwise.log 2>&1
cat $O/small_values.log $O/hconv_*.json; tail -n 3 $O/pointwise.log
