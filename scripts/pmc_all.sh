#!/bin/bash
# Counter passes of one bench configuration, each --pmc set in a run of its own (never combined with trace flags):
#   three SQ sets (issue, waits, LDS/VMEM mix), FETCH_SIZE, WRITE_SIZE, and (round 6) three sets of the CU's vector-memory path: texture-address unit (TA), data-return
#   unit (TD), L1 tag lookups and requests to L2 (TCP) — the combinations scripts/pmc_mem.sh found rocprofv3 to accept (at most four counters of one block per set).
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
           "FETCH_SIZE" "WRITE_SIZE" \
           "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d $OUT/set$i -- $BENCH > $OUT/set$i.json 2> $OUT/set$i.err || echo "set $i failed"
  echo "pmc $TAG set $i done"
done
python3 $R/scripts/pmc_summary.py $OUT $CFG $FPS > $OUT/summary.txt
cat $OUT/summary.txt
# raw per-dispatch csv files are large: keep the summaries only
rm -rf $OUT/set*/
