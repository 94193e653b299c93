#!/bin/bash
# api fuzz under the terminate-backtrace helper (tools/terminate_trace.cpp): where does an uncaught native exception come from?   fuzz_trace.sh <tag> <first seed> <count>
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-fuzztrace}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
g++ -O1 -fPIC -shared -o build/terminate_trace.so tools/terminate_trace.cpp || exit 1
LD_PRELOAD=$R/build/terminate_trace.so API_FUZZ_TRACE=$O/ops.txt timeout -k 10 900 python3 -X faulthandler scripts/api_fuzz.py ${2:-101351} ${3:-60} > $O/out.txt 2>&1; echo "rc=$?"
tail -60 $O/out.txt
tail -5 $O/ops.txt
