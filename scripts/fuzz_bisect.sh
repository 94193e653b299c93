#!/bin/bash
# diagnosis of the host heap corruption the api fuzz found: glibc's malloc checks (libc_malloc_debug + MALLOC_CHECK_=3: a trailing canary per block, checked when the block is
# freed) and a native backtrace at abort.   fuzz_bisect.sh <tag> <first> <count>
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-fuzzbisect}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
g++ -O1 -fPIC -shared -o build/terminate_trace.so tools/terminate_trace.cpp || exit 1
export LD_PRELOAD="/lib/x86_64-linux-gnu/libc_malloc_debug.so.0 $R/build/terminate_trace.so" MALLOC_CHECK_=3 PT_TRACE_ABORT=1
API_FUZZ_TRACE=$O/ops.txt API_FUZZ_NO_WRITE=1 timeout -k 10 800 python3 -X faulthandler scripts/api_fuzz.py ${2:-100001} ${3:-3000} > $O/out.txt 2>&1; echo "rc=$?"; grep -v "seeds, 0 bad" $O/out.txt | head -70 | cut -c1-220
tail -40 $O/ops.txt > $O/ops_tail.txt; tail -2 $O/ops_tail.txt
