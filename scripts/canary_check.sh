#!/bin/bash
# under the checking allocator (tools/canary_malloc.cpp: write past the end / write after free of any heap block, with the allocating library): the api fuzz with its bursts of
# one-frame submissions, the drop-in frame loop at full size and a short bench.   canary_check.sh <tag> [api-fuzz count]
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-canary}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
g++ -O1 -fPIC -shared -o build/canary_malloc.so tools/canary_malloc.cpp -ldl -lpthread || exit 1
export LD_PRELOAD=$R/build/canary_malloc.so GPU_MAX_HW_QUEUES=8
timeout -k 10 900 python3 scripts/api_fuzz.py 100001 ${2:-2000} > $O/api_fuzz.txt 2>&1; echo "api fuzz rc=$? looks $(grep -c 'seeds, 0 bad' $O/api_fuzz.txt)"; grep -a "canary_malloc" $O/api_fuzz.txt | head -2 | cut -c1-200; tail -1 $O/api_fuzz.txt
FRAMES=128 timeout -k 10 300 python3 scripts/frame_loop.py 2 0 > $O/frame_loop.txt 2>&1; echo "frame loop rc=$?"; grep -a "canary_malloc\|Msamples\|identical" $O/frame_loop.txt | cut -c1-160
timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-roofline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; grep -a "canary_malloc" $O/bench.err | head -2; cut -c1-200 $O/bench.json
