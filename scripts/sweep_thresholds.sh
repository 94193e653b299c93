#!/bin/bash
# one-at-a-time sweep of the three phase thresholds of k_extend_persist around their defaults (bench.py, two streams)
cd ${GRAFT_REPO_ROOT:-.}
run() { timeout -k 10 200 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$*', '->', d['value'], 'Ms/s  ms/step', d['ms_per_step'])
"; }
for cfg in "--config C3" "--config C4 --frames-per-step 16"; do
  run $cfg
  for v in 8 16 32 40; do run $cfg --refill-min $v; done
  for v in 2 4 16 24; do run $cfg --none-min $v; done
  for v in 4 5 7; do run $cfg --inner-keep $v; done
  run $cfg
done
