#!/bin/bash
# bench lines of the other BASELINE configurations on the round-3 build (GPU box, repo root)
O=gpurun_out/r03cfg; mkdir -p $O
python3 bench.py --mode infer --batch 128 --steps 5 --warmup 2 > $O/bench_infer128.json 2> $O/infer.err
python3 bench.py --height 240 --width 320 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_240x320.json 2> $O/240.err
python3 bench.py --task coco --batch 16 --T 6 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_coco_b16.json 2> $O/coco.err
python3 bench.py --task osie --arch resnet18 --T 8 --batch 4 --height 240 --width 320 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_osie_r18.json 2> $O/osie.err
for f in infer128 240x320 coco_b16 osie_r18; do python3 -c "import json; d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['unit'], d['ms_per_step'])"; done
