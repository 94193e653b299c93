#!/bin/bash
# round 5: (1) always-enter variant of the lazy next-BVH step on C6; (2) cycles per trip against waves per SIMD (what saturates?)
# libraries: scripts/build_variant.py enter -DLAZY_ENTER=1; cull3; prof -DPT_ASM_DEBUG -DPT_ASM_PROF --hip -DPT_ASM_DEBUG
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-r05e}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
PT_HIP_LIB=$R/build/ab/enter.so timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -k "equals_compiled or root_cull or render_parity_handwritten" > $O/parity_enter.txt 2>&1; rc=$?
echo "parity(enter) rc=$rc $(tail -1 $O/parity_enter.txt)"
if [ $rc = 0 ]; then bash scripts/ab.sh -r 2 -c "C6" -t cull3 enter 2>&1 | tee $O/ab_c6_enter.txt; fi
for cfg in C3 C6; do for b in 1 2 4 8; do
  echo "== $cfg asm_tpb=256 extend_blocks_per_cu=$b"
  PT_HIP_LIB=$R/build/ab/prof.so PT_ASM_DEBUG=1 timeout -k 10 200 python3 scripts/asm_prof.py $cfg 4 asm_tpb=256 extend_blocks_per_cu=$b 2>&1 | grep -v amdgpu.ids
done; done | tee $O/stamps_vs_occupancy.txt
