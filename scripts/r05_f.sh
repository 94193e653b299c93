#!/bin/bash
# round 5: (1) always-enter variant on C6 (perf only; it is unsound for groups with infinite boxes), (2) unstamped launch time against blocks per CU, one stream
# libraries: scripts/build_variant.py cull3; enter -DLAZY_ENTER=1
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-r05f}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
bash scripts/ab.sh -r 2 -c "C6" -t cull3 enter 2>&1 | tee $O/ab_c6_enter.txt
for cfg in C3 C6; do for b in 1 2 4 6 8; do
  bash scripts/ab.sh -r 1 -c "$cfg" -t -f "--streams 1 --asm-tpb 256 --extend-blocks-per-cu $b --steps 3" cull3
done; done 2>&1 | tee $O/launch_vs_blocks_per_cu.txt
