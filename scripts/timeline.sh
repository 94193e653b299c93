#!/bin/bash
# kernel-trace timeline of one batch (scripts/timeline.py): timeline.sh <name> [bench args...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$N -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline "$@" > $R/gpurun_out/$N.json 2> $R/gpurun_out/$N.err && python3 $R/scripts/timeline.py $R/gpurun_out/$N > $R/gpurun_out/$N.txt
rm -rf $R/gpurun_out/$N
cat $R/gpurun_out/$N.txt
