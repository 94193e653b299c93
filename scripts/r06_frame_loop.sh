#!/bin/bash
# round 6 (d): the drop-in call pattern (one draw per frame) — where the gap to batches comes from
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-r06d}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
export GPU_MAX_HW_QUEUES=8
for slots in 0 8388608 4194304; do
  echo "== streams 2 path_slots $slots FRAMES 64"; FRAMES=64 timeout -k 10 300 python3 scripts/frame_loop.py 2 $slots 2>&1 | grep -v amdgpu.ids
done | tee $O/frame_loop.txt
echo "== streams 2 path_slots 0 FRAMES 256" | tee -a $O/frame_loop.txt; FRAMES=256 timeout -k 10 300 python3 scripts/frame_loop.py 2 0 2>&1 | grep -v amdgpu.ids | tee -a $O/frame_loop.txt
echo "== streams 1 path_slots 0 FRAMES 64" | tee -a $O/frame_loop.txt; FRAMES=64 timeout -k 10 300 python3 scripts/frame_loop.py 1 0 2>&1 | grep -v amdgpu.ids | tee -a $O/frame_loop.txt
