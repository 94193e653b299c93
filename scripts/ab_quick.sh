cd $GRAFT_REPO_ROOT
A=$1; B=$2
one() { lib=$1; shift
  PT_HIP_LIB=$GRAFT_REPO_ROOT/build/ab/$lib.so timeout -k 10 300 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$lib $* ->', d['value'], 'Ms/s  ms/step', d['ms_per_step'])
"; }
PT_HIP_LIB=$GRAFT_REPO_ROOT/build/ab/$B.so timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -k "render_parity or random_scenes or n2_direct or intersect_parity or nan_slab" 2>&1 | tail -1
for r in 1 2 3; do for lib in $A $B; do one $lib; done; done
for lib in $A $B; do one $lib --streams 1; one $lib --config C4 --frames-per-step 16; one $lib --config C5 --frames-per-step 8; one $lib --config C2; done
