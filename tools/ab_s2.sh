export SP_LIBRARY=timing      # knobs below exist in libscanpaths_amd_timing.so only (make -C scanpaths_amd/csrc timing)
# short-K pointwise kernel (s2, now with float4 epilogue stores) against h2_kernel: micro-benchmark + encoder + correctness
mkdir -p gpurun_out/r03t
timeout 300 python -m pytest tests/test_ops_gpu.py -q -k "epilogue_writes" > gpurun_out/r03t/ops.log 2>&1
python3 tools/bench_pointwise.py > gpurun_out/r03t/pw_h2.log 2>&1
SP_S2=1 python3 tools/bench_pointwise.py > gpurun_out/r03t/pw_s2.log 2>&1
python3 tools/bench_backbone.py > gpurun_out/r03t/backbone_h2.json 2>/dev/null
SP_S2=1 python3 tools/bench_backbone.py > gpurun_out/r03t/backbone_s2.json 2>/dev/null
