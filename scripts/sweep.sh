#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('$*', '->', d['value'], 'Ms/s  ms/step', d['ms_per_step'], 'ext_ms', r.get('avg_launch_ms'), 'shade_ms', r.get('shade_avg_launch_ms'), 'launches', r.get('launches'))
"; }
S="--rehearse-shard 0 8"
run $S
run $S --extend-blocks-per-cu 2
run $S --extend-blocks-per-cu 3
run $S --extend-tpb 256
run $S --extend-tpb 256 --extend-blocks-per-cu 6
run $S --refill-min 8
run $S --refill-min 40
run $S --path-slots 1572864
run $S --path-slots 2097152
run $S --path-slots 2097152 --extend-blocks-per-cu 3
run $S --path-slots 3145728
