#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06h; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
for n in 64 256; do echo "== streams 2 FRAMES $n"; FRAMES=$n timeout -k 10 300 python3 scripts/frame_loop.py 2 0 2>&1 | grep -v amdgpu.ids; done | tee $O/frame_loop.txt
for per in 1 2 4; do timeout -k 10 300 python3 scripts/frame_mode.py async1 2 256 $per 2>&1 | tail -1; done | tee $O/frames_per_call.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -q -x -k "overlapped or write_frame or batch_equals or full_size or bench_default or sharding" 2>&1 | tail -3 | tee $O/pytest.txt
timeout -k 10 600 python3 scripts/api_fuzz.py 91001 300 2>&1 | tail -1 | tee $O/api_fuzz.txt
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-260 | tee $O/bench.txt
