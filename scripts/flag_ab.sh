#!/bin/bash
# A/B of bench.py flag sets with the in-tree library on ONE box, interleaved rounds (boxes of the pool differ by up to 8 %).
#   flag_ab.sh [-r ROUNDS] [-c "C3 C4"] [-t] [-s STEPS] "<flags A>" "<flags B>" ...      ("" = the defaults)   -t keeps the per-kernel HIP-event timing
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROUNDS=2; CFGS="C3"; TIMES=0; STEPS=6
while getopts ":r:c:ts:" o; do case $o in r) ROUNDS=$OPTARG;; c) CFGS=$OPTARG;; t) TIMES=1;; s) STEPS=$OPTARG;; esac; done
shift $((OPTIND-1))
cd $R
NR=""; [ $TIMES = 0 ] && NR="--no-roofline"
for round in $(seq 1 $ROUNDS); do for cfg in $CFGS; do for FL in "$@"; do
  fps=""; [ $cfg != C3 ] && fps="--frames-per-step 16"
  timeout -k 10 300 python3 bench.py --config $cfg --steps $STEPS --warmup 2 --no-cpu-baseline --no-alone-pass $NR $fps $FL 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('$cfg [$FL] ->', d['value'], 'Ms/s  ms/step', d['ms_per_step'], ('  extend avg %s  shade avg %s' % (r.get('avg_launch_ms'), r.get('shade',{}).get('avg_launch_ms'))) if r else '', flush=True)
" || echo "$cfg [$FL] FAILED"
done; done; done
