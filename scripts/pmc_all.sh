#!/bin/bash
# Counter passes of one bench configuration, each --pmc set in a run of its own (never combined with trace flags):
#   three SQ sets (issue, waits, LDS/VMEM mix), FETCH_SIZE, WRITE_SIZE.
# usage: scripts/pmc_all.sh <out-tag> <config> [extra bench.py flags]     -> gpurun_out/<out-tag>/{set*.csv dirs, summary.json, summary.txt}
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-pmc}; CFG=${2:-C3}
shift; shift
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
FPS=${FPS:-8}
BENCH="python3 $R/bench.py --config $CFG --steps 1 --warmup 0 --frames-per-step $FPS --no-cpu-baseline --no-roofline $*"
# the same run un-profiled, with statistics: segments per sample, per-kernel launch times
python3 $R/bench.py --config $CFG --steps 1 --warmup 0 --frames-per-step $FPS --no-cpu-baseline "$@" > $OUT/plain.json 2> $OUT/plain.err || echo "plain run failed"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_VMEM_WR SQ_WAVES_EQ_64 SQ_INST_LEVEL_LDS" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/set$i -- $BENCH > $OUT/set$i.json 2> $OUT/set$i.err || echo "set $i failed"
  echo "pmc $TAG set $i done"
done
python3 $R/scripts/pmc_summary.py $OUT $CFG $FPS > $OUT/summary.txt
cat $OUT/summary.txt
# raw per-dispatch csv files are large: keep the summaries only
rm -rf $OUT/set*/
