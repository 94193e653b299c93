#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-noreap}; mkdir -p $O; cd $R
g++ -O1 -fPIC -shared -o build/canary_malloc.so tools/canary_malloc.cpp -ldl -lpthread || exit 1
PT_NO_REAP=1 LD_PRELOAD=$R/build/canary_malloc.so timeout -k 10 900 python3 scripts/api_fuzz.py 100001 2500 > $O/api_fuzz.txt 2>&1; echo "api fuzz (no hipStreamQuery, pooled streams) rc=$? looks $(grep -c 'seeds, 0 bad' $O/api_fuzz.txt)"; grep -a "canary_malloc" $O/api_fuzz.txt | head -2 | cut -c1-200; tail -1 $O/api_fuzz.txt
