#!/bin/bash
# round 5, first call: the GPU suite on the round-4 kernels, the drop-in call pattern (one draw per frame) and the C3 / C4 / C6 lines of this box
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05a; mkdir -p $O; cd $R
timeout -k 10 500 python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$? $(tail -1 $O/gpu_tests.txt)"
GPU_MAX_HW_QUEUES=8 timeout -k 10 200 python3 scripts/frame_loop.py 2 > $O/frame_loop_two_streams.txt 2>&1; echo "frame loop 2 rc=$?"; cat $O/frame_loop_two_streams.txt
timeout -k 10 200 python3 scripts/frame_loop.py 1 > $O/frame_loop_one_stream.txt 2>&1; echo "frame loop 1 rc=$?"; cat $O/frame_loop_one_stream.txt
for cfg in C3 C4 C6; do fps=""; [ $cfg != C3 ] && fps="--frames-per-step 16"
  timeout -k 10 300 python3 bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline $fps > $O/bench_$cfg.json 2> $O/bench_$cfg.err; echo "bench $cfg rc=$?"
  python3 -c "
import json
for l in open('$O/bench_$cfg.json'):
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{}); print('$cfg', d['value'], 'Ms/s', d['ms_per_step'], 'extend', r.get('avg_launch_ms'), 'shade', (r.get('shade') or {}).get('avg_launch_ms'))
"; done
