#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-fuzzbisect}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
shift
for mode in "$@"; do
  PT_SCHED_DEBUG=$mode API_FUZZ_NO_WRITE=1 timeout -k 10 400 python3 -X faulthandler scripts/api_fuzz.py 100001 3000 > $O/mode$mode.txt 2>&1; echo "mode $mode rc=$? looks $(grep -c 'seeds, 0 bad' $O/mode$mode.txt)"; grep -v "seeds, 0 bad" $O/mode$mode.txt | sed -n 2,2p | cut -c1-160
done
