#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --rehearse-shard 0 8 "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$*', '->', d['value'], 'Ms/s  ms/step', d['ms_per_step'])
"; }
run
run --simple-below 65536
run --simple-below 262144
run --simple-below 786432
run --extend-cache 0
run --extend-cache 4096
run
