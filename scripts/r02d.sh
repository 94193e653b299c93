#!/bin/bash
# round-2 fourth call: A/B of kernel variants (build/ab/*.so) and option sweeps
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02d
mkdir -p $O
cd $R
one() { # lib flags...
  lib=$1; shift
  PT_HIP_LIB=$R/build/ab/$lib.so timeout -k 10 300 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$lib $* ->', d['value'], 'Ms/s  ms/step', d['ms_per_step'], ' extend avg/med', r['avg_launch_ms'], r['median_launch_ms'], ' shade avg', r['shade']['avg_launch_ms'])
"
}
for lib in B C D; do
  PT_HIP_LIB=$R/build/ab/$lib.so timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -k "render_parity or random_scenes or n2_direct or stack_widths or mapped" > $O/parity_$lib.txt 2>&1; echo "parity $lib rc=$? $(tail -1 $O/parity_$lib.txt)"
done
for round in 1 2; do for lib in A B C D; do one $lib; done; done | tee $O/ab.txt
for cfg in C4 C5; do for lib in A B C; do one $lib --config $cfg --frames-per-step 16; done; done | tee $O/ab_c45.txt
{ one B --extend-cache 0; one B --extend-cache 8192; one B --refill-min 16; one B --refill-min 20; one B --none-min 4; one B --none-min 12; one B --inner-keep 5; one B --inner-keep 7
  one B --path-slots 16777216; one B --config C4 --frames-per-step 16 --extend-cache 0; one B --config C4 --frames-per-step 16 --refill-min 16; } | tee $O/ab_opts.txt
