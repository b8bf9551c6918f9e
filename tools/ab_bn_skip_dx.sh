# same-box A/B: BatchNorm backward leaves the fp32 gradient unwritten when the producing conv reads only its split form
mkdir -p gpurun_out/r03o
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -q -x -k "train_step or T16 or gradients or fusions or reproducible or bn_act or conv2d" > gpurun_out/r03o/tests.log 2>&1
for rep in 1 2; do
python3 bench.py --steps 15 --warmup 5 --no-cpu-baseline > gpurun_out/r03o/a_skip_$rep.json 2>/dev/null
SP_BN_SKIP_DX=0 python3 bench.py --steps 15 --warmup 5 --no-cpu-baseline > gpurun_out/r03o/b_noskip_$rep.json 2>/dev/null
done
python3 tools/bench_backbone.py > gpurun_out/r03o/backbone.json 2>/dev/null
SP_BN_SKIP_DX=0 python3 tools/bench_backbone.py > gpurun_out/r03o/backbone_noskip.json 2>/dev/null
