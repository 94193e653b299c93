#!/bin/bash
# A/B of prebuilt libraries with per-kernel launch times: ab_times.sh lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
 for lib in "$@"; do
  PT_HIP_LIB=$R/build/ab/$lib timeout -k 10 200 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$lib ->', d['value'], 'Ms/s  ms/step', d['ms_per_step'], ' extend avg/med', r['avg_launch_ms'], r['median_launch_ms'], ' shade avg/med', r['shade_avg_launch_ms'], r['shade_median_launch_ms'])
"
 done
done
