#!/bin/bash
# A/B of prebuilt libraries (build/ab/*.so) on ONE box, interleaved rounds, untimed kernels: full frame and one shard of 8
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROUNDS=${ROUNDS:-2}
run() { PT_HIP_LIB=$R/build/ab/$1 timeout -k 10 120 python3 $R/bench.py --steps $2 --warmup 1 --no-cpu-baseline --no-roofline $3 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1 [$3] ->', d['value'], 'Ms/s  ms/step', d['ms_per_step'])
"; }
for round in $(seq 1 $ROUNDS); do
 for lib in "$@"; do
  run $lib 3 ""
  run $lib 8 "--rehearse-shard 0 8"
 done
done
