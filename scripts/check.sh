#!/bin/bash
# GPU suite + smoke + the driver's bench command (usage: check.sh <tag>)
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-check}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
timeout -k 10 1000 python3 -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.txt
[ $rc = 0 ] || exit 1
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_c3.json 2> $O/bench_default_c3.err; echo "bench rc=$?"; cut -c1-300 $O/bench_default_c3.json
