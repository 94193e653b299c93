cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02s
timeout -k 10 300 python3 -m pytest tests/test_gpu_multi.py -m gpu -q -k "scheduler_paths or equals_single" 2>&1 | tail -3
# soak: 150 steps = 5 G jobs per stream pair: the streams' 31-bit job counters wrap (a stream restart inside the timed region), rings and image ring wrap many times
timeout -k 10 900 python3 bench.py --steps 150 --warmup 2 > gpurun_out/r02s/soak.json 2> gpurun_out/r02s/soak.err; echo "soak rc=$?"
python3 -c "
import json
for l in open('gpurun_out/r02s/soak.json'):
    if l.startswith('{'):
        d=json.loads(l); print('soak', d['value'], 'Ms/s', d['ms_per_step'], 'ms/step', d['parity'])
"
python3 scripts/frame_loop.py 1 0 2>/dev/null | head -3
