#!/bin/bash
# round 5: the rays of a block — one contiguous range (inter0) against chunks of 64 dealt round-robin to the blocks (inter1)
# libraries: scripts/build_variant.py inter1; inter0 -DINTERLEAVE=0
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-r05k}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
bash scripts/ab.sh -p -r 2 -c "C3 C4 C5 C6" -t inter0 inter1 2>&1 | tee $O/ab_interleave.txt
bash scripts/ab.sh -r 2 -c "C3 C4" -t -f "--streams 1" inter0 inter1 2>&1 | tee -a $O/ab_interleave.txt
for n in 8; do bash scripts/ab.sh -r 2 -c "C3" -f "--rehearse-shard 0 $n --steps 8" inter0 inter1; done 2>&1 | tee -a $O/ab_interleave.txt
