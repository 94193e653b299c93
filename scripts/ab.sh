#!/bin/bash
# A/B of prebuilt libraries (build/ab/<name>.so, made by scripts/build_ab.sh) on ONE box, interleaved rounds.
#   ab.sh [-r ROUNDS] [-c "C3 C4"] [-f "<bench.py flags>"] [-t] [-p] name1 name2 ...
#     -c  configurations (default C3)          -f  extra bench.py flags, e.g. "--streams 1" or "--rehearse-shard 0 8"
#     -t  keep per-kernel HIP-event timing (prints extend / shade mean launch)     -p  first run the parity subset with the LAST library
set -o pipefail            # the parity gate below is a pipeline: without this its status would be tail's
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROUNDS=3; CFGS="C3"; FL=""; TIMES=0; PARITY=0
while getopts "r:c:f:tp" o; do case $o in r) ROUNDS=$OPTARG;; c) CFGS=$OPTARG;; f) FL=$OPTARG;; t) TIMES=1;; p) PARITY=1;; esac; done
shift $((OPTIND-1))
cd $R
if [ $PARITY = 1 ]; then
  last=${@: -1}
  PT_HIP_LIB=$R/build/ab/$last.so timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -x -q -k "render_parity or random_scenes or n2_direct or intersect_parity or nan_slab or equals_compiled" 2>&1 | tail -1 || { echo "parity subset FAILED with $last: no A/B numbers"; exit 1; }
fi
NR=""; [ $TIMES = 0 ] && NR="--no-roofline"
for round in $(seq 1 $ROUNDS); do for cfg in $CFGS; do for lib in "$@"; do
  fps=""; [ $cfg != C3 ] && fps="--frames-per-step 16"
  PT_HIP_LIB=$R/build/ab/$lib.so timeout -k 10 300 python3 bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline $NR $fps $FL 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('$cfg $lib [$FL] ->', d['value'], 'Ms/s  ms/step', d['ms_per_step'], ('  extend avg %s  shade avg %s' % (r.get('avg_launch_ms'), r.get('shade',{}).get('avg_launch_ms'))) if r else '')
"
done; done; done
