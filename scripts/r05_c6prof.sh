#!/bin/bash
# round 5: where C6 spends its time after the object-loop cull — cycle stamps of the fused trip (developer build) and the counter passes
# library: scripts/build_variant.py prof -DPT_ASM_DEBUG -DPT_ASM_PROF --hip -DPT_ASM_DEBUG
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-r05d}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
for cfg in C6 C4 C3; do PT_HIP_LIB=$R/build/ab/prof.so PT_ASM_DEBUG=1 timeout -k 10 200 python3 scripts/asm_prof.py $cfg 4 > $O/stamps_$cfg.txt 2>&1; echo "stamps $cfg rc=$?"; grep -v amdgpu.ids $O/stamps_$cfg.txt; done
FPS=4 bash scripts/pmc_all.sh $T/pmc_C6 C6 > $O/pmc_C6.log 2>&1; echo "pmc C6 rc=$?"; tail -40 $O/pmc_C6/summary.txt
