#!/bin/bash
# one-at-a-time option sweep of bench.py on one box: sweep_opts.sh "<common flags>" "<flag set 1>" "<flag set 2>" ...   (each twice, interleaved)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
COMMON=$1; shift
for round in 1 2; do for fl in "$@"; do
  timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline $COMMON $fl 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('[$COMMON] [$fl] ->', d['value'], 'Ms/s  ms/step', d['ms_per_step'], ('  extend avg %s  shade avg %s' % (r.get('avg_launch_ms'), r.get('shade',{}).get('avg_launch_ms'))) if r else '')
"
done; done
