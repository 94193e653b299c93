#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06j; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
for n in 64 64 256; do echo "== streams 2 FRAMES $n"; FRAMES=$n timeout -k 10 300 python3 scripts/frame_loop.py 2 0 2>&1 | grep -v amdgpu.ids; done | tee $O/frame_loop.txt
echo "== streams 1 FRAMES 64" | tee -a $O/frame_loop.txt; FRAMES=64 timeout -k 10 300 python3 scripts/frame_loop.py 1 0 2>&1 | grep -v amdgpu.ids | tee -a $O/frame_loop.txt
