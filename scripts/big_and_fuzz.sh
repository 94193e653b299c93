#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-big}; mkdir -p $O; cd $R
timeout -k 10 300 python3 scripts/big_scene.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/big.txt
timeout -k 10 500 python3 scripts/big_scene.py 1415 1415 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/big.txt
timeout -k 10 900 python3 scripts/fuzz_campaign.py 80001 1500 2>&1 | tail -2 | tee $O/fuzz.txt
