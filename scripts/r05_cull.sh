#!/bin/bash
# round 5: the per-ray cull of the object loop — parity first, then C6 A/B against the round-4 library on ONE box
# libraries: scripts/build_variant.py cull3; cull1 -DLAZY_CAP=1; cull2 -DLAZY_CAP=2; build/ab/r4.so = libpt_hip.so built from the round-4 tree (git worktree at afac731)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r05b}; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "equals_compiled or root_cull or render_parity_handwritten or intersect_parity or random_scenes or full_size_properties" > $O/parity.txt 2>&1; rc=$?
echo "parity rc=$rc $(tail -1 $O/parity.txt)"
[ $rc = 0 ] || { tail -30 $O/parity.txt; exit 1; }
bash scripts/ab.sh -r 2 -c "C6" -t r4 cull3 cull1 cull2 2>&1 | tee $O/ab_c6.txt
bash scripts/ab.sh -r 1 -c "C6" -t -f "--asm-root-cull 0" cull3 2>&1 | tee -a $O/ab_c6.txt
bash scripts/ab.sh -r 1 -c "C3 C4 C1" r4 cull3 2>&1 | tee $O/ab_others.txt
