#!/bin/bash
# the compiled (1) and the hand-written (2) intersect kernel side by side on one box: bench_modes.sh "<configs>" "<extra bench flags>" [rounds]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
CFGS=${1:-C3}; FL=$2; ROUNDS=${3:-2}
for round in $(seq 1 $ROUNDS); do for cfg in $CFGS; do for mode in 1 2; do
  fps=""; [ $cfg != C3 ] && fps="--frames-per-step 16"
  timeout -k 10 300 python3 bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --extend-mode $mode $fps $FL 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('$cfg mode $mode [$FL] ->', d['value'], 'Ms/s  ms/step', d['ms_per_step'], ('  extend avg %s  shade avg %s' % (r.get('avg_launch_ms'), r.get('shade',{}).get('avg_launch_ms'))) if r else '')
"
done; done; done
