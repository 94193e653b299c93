#!/bin/bash
# round 5: what the wider traversal-stack entries cost where the narrower ones suffice (library: scripts/build_variant.py cur)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-r05x}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
for f in "" "--stack-mode 2"; do bash scripts/ab.sh -r 2 -c "C4 C3" -t -f "$f" cur; done 2>&1 | tee $O/ab_stack_width.txt
bash scripts/ab.sh -r 2 -c "C3" -t -f "--stack-mode 1" cur 2>&1 | tee -a $O/ab_stack_width.txt
