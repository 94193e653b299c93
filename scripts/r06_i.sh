#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06i; mkdir -p $O
bash scripts/r06_check.sh r06i || exit 1
# the C host on ROCm's own runtime beside the Python bench (torch's bundled runtime) on ONE box, interleaved
for k in 1 2; do
  timeout -k 10 600 python3 scripts/c_host_bench.py --steps 20 --warmup 5 --streams 2 2>&1 | grep "^{" | tee -a $O/c_host.txt
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(json.dumps({'host': 'python bench.py', 'hip_runtime': d['hip_runtime'], 'ms_per_step': d['ms_per_step'], 'value': d['value']}))" | tee -a $O/c_host.txt
done
timeout -k 10 600 python3 scripts/c_host_bench.py --steps 20 --warmup 5 --streams 1 2>&1 | grep "^{" | tee -a $O/c_host.txt
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --streams 1 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(json.dumps({'host': 'python bench.py --streams 1', 'hip_runtime': d['hip_runtime'], 'ms_per_step': d['ms_per_step'], 'value': d['value']}))" | tee -a $O/c_host.txt
