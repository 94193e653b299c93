#!/bin/bash
# A/B of prebuilt libraries (build/ab/*.so) on one box over several configs: ab3.sh "C3 C4" lib1.so lib2.so
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFGS=$1; shift
for round in ${ROUNDS:-1 2}; do
 for cfg in $CFGS; do
  for lib in "$@"; do
   PT_HIP_LIB=$R/build/ab/$lib timeout -k 10 200 python3 $R/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --frames-per-step 16 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$cfg $lib ->', d['value'], 'Ms/s  ms/step', d['ms_per_step'])
"
  done
 done
done
