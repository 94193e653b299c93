#!/bin/bash
# classification run: the api fuzz (with its bursts of one-frame submissions) on an earlier round's library (build/ab/<name>.so)   fuzz_old_lib.sh <tag> <lib> <first> <count>
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-fuzzold}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
g++ -O1 -fPIC -shared -o build/terminate_trace.so tools/terminate_trace.cpp || exit 1
PT_HIP_LIB=$R/build/ab/${2:-r5}.so LD_PRELOAD=$R/build/terminate_trace.so API_FUZZ_TRACE=$O/ops.txt timeout -k 10 1000 python3 -X faulthandler scripts/api_fuzz.py ${3:-100001} ${4:-4000} > $O/out.txt 2>&1; echo "rc=$?"
tail -25 $O/out.txt | cut -c1-200
