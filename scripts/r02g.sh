#!/bin/bash
# several independent wavefront streams on ONE GPU (a device listed k times in pt_create_multi): launch tails and the two kernels' resources overlap
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02g
mkdir -p $O
cd $R
run() { timeout -k 10 300 python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$*', '->', d['value'], 'Ms/s  ms/step', d['ms_per_step'])
"; }
{ run; run --devices 0,0; run --devices 0,0,0; run --devices 0,0,0,0; run --devices 0,0 --path-slots 4194304; run --devices 0,0,0,0 --path-slots 2097152; run --devices 0,0 --path-slots 16777216
  run --config C4 --frames-per-step 16; run --config C4 --frames-per-step 16 --devices 0,0; run --config C4 --frames-per-step 16 --devices 0,0,0
  run --config C5 --frames-per-step 8; run --config C5 --frames-per-step 8 --devices 0,0
  run --config C2; run --config C2 --devices 0,0; } | tee $O/streams.txt
