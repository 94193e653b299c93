#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-fuzzbisect}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
for mode in 2 1 0; do
  PT_SCHED_DEBUG=$mode API_FUZZ_NO_WRITE=1 timeout -k 10 600 python3 -X faulthandler scripts/api_fuzz.py 100001 3000 > $O/mode$mode.txt 2>&1; echo "mode $mode rc=$?"; grep -v "seeds, 0 bad" $O/mode$mode.txt | head -6 | cut -c1-200; tail -1 $O/mode$mode.txt | cut -c1-100
done
