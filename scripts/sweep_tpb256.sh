cd $GRAFT_REPO_ROOT
run() { PT_HIP_LIB=$GRAFT_REPO_ROOT/build/ab/X0.so timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$*', '->', d['value'], 'Ms/s  ms/step', d['ms_per_step'])
"; }
for c in 1024 2048 4096 8192 16384; do run --config C4 --frames-per-step 16 --extend-tpb 256 --extend-blocks-per-cu 8 --extend-cache $c; done
for c in 1024 2048 4096 8192 16384; do run --config C5 --frames-per-step 8 --extend-tpb 256 --extend-blocks-per-cu 8 --extend-cache $c; done
run --config C5 --frames-per-step 8
for c in 2048 4096 8192; do run --config C2 --extend-tpb 256 --extend-blocks-per-cu 8 --extend-cache $c; done
run --config C2
for c in 2048 4096 6144 12288; do run --extend-tpb 256 --extend-blocks-per-cu 8 --extend-cache $c; done
