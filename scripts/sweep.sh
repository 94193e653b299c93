#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('$*', '->', d['value'], 'Ms/s  ext_ms', r.get('avg_launch_ms'), 'shade_ms', r.get('shade_avg_launch_ms'), 'frac', r.get('frac'))
"; }
F="--frames-per-step 8"
run --config C3 $F
run --config C3 $F --path-slots 4194304
run --config C3 $F --path-slots 8388608
run --config C3 $F --path-slots 4194304 --extend-tpb 256 --extend-cache 0
run --config C3 $F --path-slots 4194304 --extend-tpb 1024 --extend-cache 65536
run --config C3 $F --path-slots 4194304 --refill-min 8
run --config C3 $F --path-slots 4194304 --refill-min 32
run --config C3 --path-slots 4194304
run --config C2 $F --path-slots 4194304
