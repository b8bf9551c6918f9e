cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04x; mkdir -p $O
for i in 1 2 3; do t0=$(date +%s); timeout 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-dense-leg > $O/b$i.json 2> $O/b$i.err; echo "rc=$? $(( $(date +%s) - t0 )) s"; tail -2 $O/b$i.err; python3 -c "
import json,sys; d=json.loads(open('$O/b$i.json').readline()); print(d['value'], d['ms_per_step'])
for g in d['roofline']['timed_gemms']: print('   ', g['kernel'][:34], g['M'], g['N'], g['K'], g['launches_per_step'], g['avg_launch_ms'], g['tflops'])"; done
