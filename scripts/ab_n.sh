#!/bin/bash
# bench several A/B libraries of build/ab/ in turn:  ab_n.sh "<bench flags>" lib1 lib2 ...   (three rounds)
cd $GRAFT_REPO_ROOT
FL=$1; shift
one() { lib=$1
  PT_HIP_LIB=$GRAFT_REPO_ROOT/build/ab/$lib.so timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline $FL 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$lib $FL ->', d['value'], 'Ms/s  ms/step', d['ms_per_step'])
"; }
for r in 1 2 3; do for lib in "$@"; do one $lib; done; done
