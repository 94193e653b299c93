#!/bin/bash
# round 6 (e): kernel traces of the per-frame and the batched call pattern (which launches does a submission per frame add?)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06e; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for m in async1 batch32; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$m -- python3 $R/scripts/frame_mode.py $m 2 128 > $O/$m.txt 2>&1
  tail -1 $O/$m.txt
  f=$(find $O/$m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -d, -f1-7 "$f" | cut -c1-200 > $O/${m}_kernel_stats.csv
  rm -rf $O/$m
done
cd $R
for per in 1 2 4 8; do timeout -k 10 300 python3 scripts/frame_mode.py async1 2 128 $per 2>&1 | tail -1; done | tee $O/frames_per_call.txt
