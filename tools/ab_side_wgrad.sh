# same-box A/B: deferred side-stream weight gradient of the h-gate conv on / off
mkdir -p gpurun_out/r03k
timeout 600 python -m pytest tests/test_model_gpu.py -q -x -k "train_step or T16 or gradients or fusions or reproducible" > gpurun_out/r03k/model.log 2>&1
for rep in 1 2; do
python3 bench.py --steps 15 --warmup 5 --no-cpu-baseline > gpurun_out/r03k/a_side_$rep.json 2>/dev/null
SP_SIDE_WGRAD=0 python3 bench.py --steps 15 --warmup 5 --no-cpu-baseline > gpurun_out/r03k/b_noside_$rep.json 2>/dev/null
done
