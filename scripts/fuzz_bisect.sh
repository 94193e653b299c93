#!/bin/bash
# the burst fuzz under the canary allocator (tools/canary_malloc.cpp): WHICH heap block gets written past its end?
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-fuzzcanary}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
g++ -O1 -fPIC -shared -o build/canary_malloc.so tools/canary_malloc.cpp -ldl -lpthread && g++ -O1 -fPIC -shared -o build/terminate_trace.so tools/terminate_trace.cpp || exit 1
LD_PRELOAD="$R/build/canary_malloc.so $R/build/terminate_trace.so" PT_TRACE_ABORT=1 API_FUZZ_TRACE=$O/ops.txt API_FUZZ_NO_WRITE=1 timeout -k 10 900 python3 scripts/api_fuzz.py 100001 ${2:-4000} > $O/out.txt 2>&1; echo "rc=$? looks $(grep -c 'seeds, 0 bad' $O/out.txt)"; grep -v "seeds, 0 bad" $O/out.txt | sed -n 2,60p | cut -c1-220
tail -60 $O/ops.txt > $O/ops_tail.txt
