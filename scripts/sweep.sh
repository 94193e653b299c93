#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('$*', '->', d['value'], 'Ms/s  ms/step', d['ms_per_step'], 'ext avg/med', r.get('avg_launch_ms'), r.get('median_launch_ms'), 'shade', r.get('shade_avg_launch_ms'), 'frac', r.get('frac'))
"; }
run --dual-pool 0
run --dual-pool 1
run --dual-pool 0
run --dual-pool 1
run --dual-pool 1 --path-slots 8388608
run --dual-pool 1 --rehearse-shard 0 8
run --dual-pool 0 --rehearse-shard 0 8
