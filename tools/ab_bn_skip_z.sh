# same-box A/B: inner BatchNorms leave their fp32 output unwritten when the consumer conv runs from the split operand
mkdir -p gpurun_out/r03p
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -q -x -k "train_step or T16 or gradients or fusions or reproducible or bn_act or conv2d or checkpoint" > gpurun_out/r03p/tests.log 2>&1
for rep in 1 2; do
python3 bench.py --steps 15 --warmup 5 --no-cpu-baseline > gpurun_out/r03p/a_skipz_$rep.json 2>/dev/null
SP_BN_SKIP_Z=0 python3 bench.py --steps 15 --warmup 5 --no-cpu-baseline > gpurun_out/r03p/b_noskipz_$rep.json 2>/dev/null
done
python3 tools/bench_backbone.py > gpurun_out/r03p/backbone.json 2>/dev/null
SP_BN_SKIP_Z=0 python3 tools/bench_backbone.py > gpurun_out/r03p/backbone_noskipz.json 2>/dev/null
