#!/bin/bash
# round-2 third call: full GPU suite, then cur vs flat intersect kernel, C4 stack widths / node layouts, pool sizes
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02c
mkdir -p $O
cd $R
timeout -k 10 1000 python3 -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.txt
one() { # lib flags...
  lib=$1; shift
  PT_HIP_LIB=$R/build/ab/$lib.so timeout -k 10 300 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$lib $* ->', d['value'], 'Ms/s  ms/step', d['ms_per_step'], ' extend avg/med', r['avg_launch_ms'], r['median_launch_ms'], ' shade avg', r['shade']['avg_launch_ms'])
"
}
PT_HIP_LIB=$R/build/ab/flat.so timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -k "render_parity or random_scenes or n2_direct or stack_widths" > $O/parity_flat.txt 2>&1; echo "parity flat rc=$? $(tail -1 $O/parity_flat.txt)"
for round in 1 2; do for lib in cur flat; do one $lib; done; done | tee $O/ab.txt
for lib in cur flat; do one $lib --path-slots 16777216; done | tee -a $O/ab.txt
for cfg in C4 C5; do for lib in cur flat; do one $lib --config $cfg --frames-per-step 16; done; done | tee $O/ab_c45.txt
{ one cur --config C4 --frames-per-step 16 --stack-mode 2
  one cur --config C4 --frames-per-step 16 --bfs-nodes 300
  one cur --config C4 --frames-per-step 16 --bfs-nodes 4200
  one cur --config C4 --frames-per-step 16 --bfs-nodes 33000
  one cur --config C4 --frames-per-step 16 --path-slots 16777216
  one cur --config C4 --frames-per-step 16 --extend-cache 8192
  one cur --config C5 --frames-per-step 16 --path-slots 16777216
  one cur --config C5 --frames-per-step 16 --bfs-nodes 300
  one cur --bfs-nodes 300
  one cur --config C2 --frames-per-step 8; } | tee $O/ab_c4.txt
