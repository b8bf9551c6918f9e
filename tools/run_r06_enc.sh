#!/bin/bash
# Round 6: encoder census + backbone (after the narrow weight-gradient change), then the HBM-side counter passes
O=gpurun_out/r06f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 tools/encoder_census.py > $O/census.log 2> $O/census.err
python3 tools/bench_backbone.py --reps 5 > $O/backbone.json 2> $O/backbone.err
head -30 $O/census.log; cat $O/backbone.json
bash tools/run_r06_pmc_hbm.sh > $O/pmc_hbm.log 2>&1
tail -40 $O/pmc_hbm.log
