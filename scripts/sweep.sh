#!/bin/bash
# one-GPU rehearsal sweeps of a single shard (bench.py --rehearse-shard): prints value and ms/step per option set
R=${GRAFT_REPO_ROOT:-$(pwd)}
SH=${SHARD:-"0 8"}
run() { timeout -k 10 120 python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --rehearse-shard $SH "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$SH | $*', '->', d['value'], 'Ms/s  ms/step', d['ms_per_step'])
"; }
for a in "$@"; do run $a; done
