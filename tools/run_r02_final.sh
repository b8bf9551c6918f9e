#!/bin/bash
# Round-2 final artifact run (on the GPU box, from the repo root): everything lands in gpurun_out/r02final/ and is copied into
# profiles/ by hand.  Separate PMC passes, no trace domains combined with counters.
O=gpurun_out/r02final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
i=0
for ctr in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_$i -o p -- python3 tools/bench_hconv.py > $O/pmc_$i.log 2>&1
  f=$(find $O/pmc_$i -name "p_counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/pmc_$i/p_counter_collection.csv
done
python3 tools/parse_pmc.py $O/pmc_ 6 $O/r02_pmc_hconv.json > $O/parse_pmc.log 2>&1
find $O -name "p_counter_collection.csv" -delete
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_backbone -o p -- python3 tools/bench_backbone.py > $O/backbone_prof.json 2> $O/backbone_prof.err
f=$(find $O/prof_backbone -name "p_kernel_trace.csv" | head -1); python3 tools/summarize_prof.py "${f%_kernel_trace.csv}" $O/r02_backbone > $O/summ_backbone.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
f=$(find $O/prof_bench -name "p_kernel_trace.csv" | head -1); python3 tools/summarize_prof.py "${f%_kernel_trace.csv}" $O/r02 > $O/summ_bench.log 2>&1
find $O -name "*trace.csv" -delete
python3 tools/bench_backbone.py > $O/backbone.json 2> $O/backbone.err
python3 bench.py --mode infer --batch 128 --steps 5 --warmup 2 > $O/bench_infer128.json 2> $O/bench_infer128.err
python3 bench.py --height 240 --width 320 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_240x320.json 2> $O/bench_240x320.err
python3 bench.py --task coco --batch 16 --T 6 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_coco_b16.json 2> $O/bench_coco.err
python3 bench.py --task osie --arch resnet18 --T 8 --batch 4 --height 240 --width 320 --steps 10 --warmup 3 --cpu-batch 4 > $O/bench_osie_r18.json 2> $O/bench_osie.err
python3 bench.py --precision f16x1 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_f16x1.json 2> $O/bench_f16x1.err
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -4 $O/pytest.log; cut -c1-300 $O/backbone.json; cut -c1-200 $O/bench_infer128.json; cut -c1-200 $O/bench.json
