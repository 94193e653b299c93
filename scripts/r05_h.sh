#!/bin/bash
# round 5: k_shade decides the end of a sample before it shades (G4 / J for ending lanes only, job pull under the loads, ONE Box-Muller site for lobes and lenses)
# libraries: scripts/build_variant.py base (the tree before the k_shade change); shade5 (after)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-r05h}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
PT_HIP_LIB=$R/build/ab/shade5.so timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; rc=$?
echo "gpu tests rc=$rc $(tail -1 $O/gpu_tests.txt)"
[ $rc = 0 ] || { tail -40 $O/gpu_tests.txt; exit 1; }
bash scripts/ab.sh -r 2 -c "C3 C4 C5 C2" -t base shade5 2>&1 | tee $O/ab_shade.txt
