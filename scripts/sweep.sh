#!/bin/bash
# quick tuning sweeps; prints value + extend/shade avg launch ms
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('$*', '->', d['value'], 'Ms/s  ext_ms', r.get('avg_launch_ms'), 'shade_ms', r.get('shade_avg_launch_ms'), 'nodes/seg', r.get('per_segment',{}).get('nodes'), 'S', r.get('segments_per_sample'))
"; }
for b in 16384 20480 32768 49152 65536; do run --config C3 --frames-per-step 8 --lds-budget $b; done
run --config C2 --frames-per-step 8
run --config C2 --frames-per-step 8 --lds-budget 0
for s in 524288 2097152 4194304; do run --config C3 --frames-per-step 8 --path-slots $s; done
