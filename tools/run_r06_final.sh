#!/bin/bash
# Round-6 artifact run (GPU box, repo root): full -m gpu suite, bench lines (headline with its dense / length-law / drop-in legs and the CPU
# leg, --dropin as the line, dense backward, RCCL world of one), kernel trace + gap / critical-path analysis, the other BASELINE configurations.
# Counter passes: tools/run_r06_pmc.sh (dominant GEMMs), tools/run_r06_pmc_hbm.sh (HBM-bound kernels); encoder: tools/run_r06_enc.sh.
O=gpurun_out/r06final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -q --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_nocpu.json 2> $O/bench_nocpu.err
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --dropin > $O/bench_dropin.json 2> $O/bench_dropin.err
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --dense-backward --no-dropin-leg > $O/bench_dense.json 2> $O/bench_dense.err
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --force-bucketer --no-dense-leg > $O/bench_bucketer.json 2> $O/bench_bucketer.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dense-leg --no-dropin-leg --no-length-leg > $O/bench_prof.json 2> $O/bench_prof.err
f=$(find $O/prof_bench -name "p_kernel_trace.csv" | head -1); python3 tools/summarize_prof.py "${f%_kernel_trace.csv}" $O/r06 > $O/summ_bench.log 2>&1
python3 tools/trace_gaps.py "$f" > $O/gaps.log 2>&1
python3 tools/trace_bwd_steps.py "$f" > $O/bwd_steps.log 2>&1
python3 tools/launch_census.py "${f%_kernel_trace.csv}_kernel_stats.csv" 9 > $O/launch_census.log 2>&1
python3 tools/aten_census.py > $O/aten_census.log 2>&1
python3 tools/bench_rank1.py > $O/bench_rank1.json 2> $O/bench_rank1.err
find $O -name "*trace.csv" -delete
python3 bench.py --steps 10 --warmup 3 --height 240 --width 320 --no-cpu-baseline > $O/bench_240x320.json 2> $O/bench_240x320.err
python3 bench.py --steps 10 --warmup 3 --task osie --arch resnet18 --T 8 --batch 4 --height 240 --width 320 --no-cpu-baseline > $O/bench_osie_r18.json 2> $O/bench_osie_r18.err
python3 bench.py --steps 10 --warmup 3 --task coco --batch 16 --T 6 --no-cpu-baseline > $O/bench_coco_b16.json 2> $O/bench_coco_b16.err
python3 bench.py --steps 5 --warmup 2 --mode infer --batch 128 > $O/bench_infer128.json 2> $O/bench_infer128.err
python3 examples/train_synthetic.py > $O/example.log 2>&1; echo "example rc=$?" >> $O/example.log
python3 tools/bench_backbone.py > $O/backbone.json 2> $O/backbone.err
python3 tools/encoder_census.py > $O/encoder_census.log 2> $O/encoder_census.err
python3 tools/bench_pointwise.py > $O/pointwise.log 2>&1
if [ -d _prev_tree ]; then bash tools/ab_prev_tree.sh "" 3 > $O/ab_prev_tree.log 2>&1; fi
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -n 4 $O/pytest.log; tail -n 3 $O/example.log; cut -c1-300 $O/bench.json
