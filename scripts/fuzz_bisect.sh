#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-fuzzcanary}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
g++ -O1 -fPIC -shared -o build/canary_malloc.so tools/canary_malloc.cpp -ldl -lpthread || exit 1
shift
for mode in "$@"; do
  PT_SCHED_DEBUG=$mode LD_PRELOAD=$R/build/canary_malloc.so API_FUZZ_NO_WRITE=1 timeout -k 10 300 python3 scripts/api_fuzz.py 100001 ${COUNT:-250} > $O/mode$mode.txt 2>&1; echo "mode $mode rc=$? looks $(grep -c 'seeds, 0 bad' $O/mode$mode.txt)"; grep -a "canary_malloc" $O/mode$mode.txt | head -2 | cut -c1-200
done
