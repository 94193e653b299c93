#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --frames-per-step 16 "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('$*', '->', d['value'], 'Ms/s  ext avg/med', r.get('avg_launch_ms'), r.get('median_launch_ms'), 'shade', r.get('shade_avg_launch_ms'))
"; }
for k in 8 7 6 5 4 2 0; do run --inner-keep $k; done
run --inner-keep 6 --refill-min 16
run --inner-keep 6 --refill-min 32
run --inner-keep 6 --extend-blocks-per-cu 4
run --inner-keep 6 --extend-cache 32768
run --inner-keep 6
