#!/bin/bash
# A/B of prebuilt libraries (build/ab/*.so) on ONE box, interleaved rounds: prints value, extend ms, shade ms
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROUNDS=${ROUNDS:-2}
for round in $(seq 1 $ROUNDS); do
 for lib in "$@"; do
  PT_HIP_LIB=$R/build/ab/$lib python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --frames-per-step 16 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('$lib round $round ->', d['value'], 'Ms/s  ext avg/med', r.get('avg_launch_ms'), r.get('median_launch_ms'), 'shade avg/med', r.get('shade_avg_launch_ms'), r.get('shade_median_launch_ms'))
"
 done
done
