#!/bin/bash
# round 5: 24-bit traversal-stack entries in the hand-written kernel (trees beyond 131071 nodes): parity, the 1 M- and 4 M-triangle meshes end to end, rates on both kernels
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-r05w}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "equals_compiled or render_parity_handwritten or intersect_parity or random_scenes or c4_big" > $O/parity.txt 2>&1; rc=$?
echo "parity rc=$rc $(tail -1 $O/parity.txt)"; [ $rc = 0 ] || { tail -30 $O/parity.txt; exit 1; }
timeout -k 10 300 python3 scripts/big_scene.py 2>&1 | grep -v amdgpu.ids | tee $O/big_scene.txt
timeout -k 10 500 python3 scripts/big_scene.py 1415 1415 2>&1 | grep -v amdgpu.ids | tee -a $O/big_scene.txt
timeout -k 10 400 python3 scripts/fallback_paths.py 2>&1 | grep -v amdgpu.ids | tee $O/fallback_paths.txt
