#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-fuzzbisect}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
g++ -O1 -fPIC -shared -o build/terminate_trace.so tools/terminate_trace.cpp || exit 1
export LD_PRELOAD=$R/build/terminate_trace.so
API_FUZZ_NO_WRITE=1 timeout -k 10 600 python3 -X faulthandler scripts/api_fuzz.py 100001 3000 > $O/no_write.txt 2>&1; echo "no_write rc=$?"; tail -3 $O/no_write.txt | cut -c1-160
API_FUZZ_NO_BURST=1 timeout -k 10 600 python3 -X faulthandler scripts/api_fuzz.py 100001 3000 > $O/no_burst.txt 2>&1; echo "no_burst rc=$?"; tail -3 $O/no_burst.txt | cut -c1-160
