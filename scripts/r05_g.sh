#!/bin/bash
# round 5: what the extra bytes of a record that carries its left child's record (80 B) or a leaf child's triangle (48 B) cost per node fetch, unused (the cost side of DESIGN.md §7's two experiments)
# libraries: scripts/build_variant.py base; vgpr80 -DFETCH_EXTRA=1; extra48 -DFETCH_EXTRA=48; extra80 -DFETCH_EXTRA=80
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-r05g}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
bash scripts/ab.sh -p -r 2 -c "C3 C4 C6" -t base vgpr80 extra48 extra80 2>&1 | tee $O/ab_fetch_extra.txt
