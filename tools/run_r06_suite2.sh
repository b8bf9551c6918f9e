#!/bin/bash
# Round 6, late: same-box A/B against the round-5 tree (_prev_tree/), then the full GPU suite for its wall time
O=gpurun_out/r06u; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/ab_prev_tree.sh "" 3 > $O/ab_r05_r06.log 2>&1; cat $O/ab_r05_r06.log
python -m pytest tests -m gpu -q --durations=25 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 30 $O/pytest.log
