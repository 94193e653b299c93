#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-fuzzbisect}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
API_FUZZ_NO_WRITE=1 timeout -k 10 500 python3 -X faulthandler scripts/api_fuzz.py 100001 ${2:-4000} > $O/stamps.txt 2>&1; echo "stamps rc=$? looks $(grep -c 'seeds, 0 bad' $O/stamps.txt)"; grep -v "seeds, 0 bad" $O/stamps.txt | sed -n 2,12p | cut -c1-160
