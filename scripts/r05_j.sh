#!/bin/bash
# round 5: 512-thread blocks (3 per CU over a 32 KB tile) when two streams share the GPU; 64-B node records on C6 after the object-loop cull
# library: scripts/build_variant.py cur
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-r05j}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -k "equals_compiled" > $O/parity.txt 2>&1; rc=$?; echo "parity rc=$rc $(tail -1 $O/parity.txt)"; [ $rc = 0 ] || { tail -30 $O/parity.txt; exit 1; }
for f in "" "--asm-tpb 512" "--asm-tpb 512 --extend-cache 24576"; do bash scripts/ab.sh -r 2 -c "C3 C4 C5 C6" -t -f "$f" cur; done 2>&1 | tee $O/ab_tpb512.txt
for f in "" "--asm-node-layout 1" "--asm-node-layout 0"; do bash scripts/ab.sh -r 2 -c "C6" -t -f "$f" cur; done 2>&1 | tee $O/ab_c6_layout.txt
