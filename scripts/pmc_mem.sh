#!/bin/bash
# Memory-pipeline counters (TA / TCP / TD) of the two kernels, one stream so that each kernel has the chip to itself.
# Each --pmc set in a run of its own, never combined with trace flags; at most four counters of one block per set (five TCP counters: rocprofv3 aborts).
# usage: scripts/pmc_mem.sh <out-tag> <config> [extra bench.py flags]   -> gpurun_out/<out-tag>/mem_summary.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-pmcmem}; CFG=${2:-C3}
shift; shift
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
FPS=${FPS:-8}
BENCH="python3 $R/bench.py --config $CFG --steps 1 --warmup 0 --frames-per-step $FPS --no-cpu-baseline --no-roofline --streams 1 $*"
i=0
for set in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum" \
           "TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_BUSY_avr TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $OUT/mem$i -- $BENCH > $OUT/mem$i.json 2> $OUT/mem$i.err || echo "set $i failed"
  echo "pmc_mem $TAG set $i done"
done
python3 - $OUT <<'PY' > $OUT/mem_summary.txt
import collections, csv, glob, re, sys
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(f"{d}/mem*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_]+|pt_extend_asm)", row["Kernel_Name"])
        k = m.group(1) if m else "other"
        a = agg[k][row["Counter_Name"]]
        a[0] += 1; a[1] += float(row["Counter_Value"])
for k in ("pt_extend_asm", "k_extend_persist", "k_shade"):
    if k not in agg: continue
    print(k)
    for c, (n, v) in sorted(agg[k].items()):
        print(f"   {c:44s} launches {n:6d}  total {v:.6g}  per launch {v / n:.6g}")
PY
cat $OUT/mem_summary.txt
rm -rf $OUT/mem*/
