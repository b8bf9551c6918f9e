export SP_LIBRARY=timing      # knobs below exist in libscanpaths_amd_timing.so only (make -C scanpaths_amd/csrc timing)
mkdir -p gpurun_out/r03j
for rep in 1 2; do
python3 bench.py --steps 15 --warmup 5 --no-cpu-baseline > gpurun_out/r03j/a_default_$rep.json 2>/dev/null
SP_LSTM_H_PLANES=0 python3 bench.py --steps 15 --warmup 5 --no-cpu-baseline > gpurun_out/r03j/b_noplanes_$rep.json 2>/dev/null
SP_LSTM_H_PLANES=0 SP_LSTM_EPI=1 python3 bench.py --steps 15 --warmup 5 --no-cpu-baseline > gpurun_out/r03j/c_noplanes_direct_$rep.json 2>/dev/null
done
