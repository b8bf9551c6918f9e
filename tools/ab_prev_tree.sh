#!/bin/bash
# Same-box, interleaved: a copy of an earlier commit (its own library built there) against this tree, optionally with one switch off.
#   rm -rf _prev_tree && mkdir _prev_tree && git archive <commit> | tar -x -C _prev_tree && make -C _prev_tree/scanpaths_amd/csrc -j8
#   gpurun -- bash tools/ab_prev_tree.sh [SP_SWITCH_TO_TURN_OFF] [rounds=3]
export SP_ALLOW_ENV_TUNING=1
VAR=$1; ROUNDS=${2:-3}
O=gpurun_out/ab_prev; mkdir -p $O
line() { grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], 'img/s', d['ms_per_step'], 'ms; dgrad', [g['avg_ms'] for g in d.get('timed_gemms', []) if 'dgrad' in g.get('name','')][:1])"; }
for r in $(seq 1 $ROUNDS); do
  (cd _prev_tree && python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-dropin-leg 2>/dev/null | line "prev round $r")
  python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-dropin-leg --no-length-leg 2>/dev/null | line "this round $r"
  if [ -n "$VAR" ]; then env $VAR=0 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-dropin-leg 2>/dev/null | line "this, $VAR=0 round $r"; fi
done | tee $O/ab.log
