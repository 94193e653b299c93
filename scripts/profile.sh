#!/bin/bash
# rocprofv3 passes for profiles/: kernel trace + stats, then HBM counters in their own runs (never combined with traces).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/bench_trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --frames-per-step 16 --no-cpu-baseline --no-roofline > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --frames-per-step 16 --no-cpu-baseline --no-roofline > $OUT/bench_write.json 2> $OUT/bench_write.err
find $OUT -name "*.csv" | head -20
