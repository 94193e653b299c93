#!/bin/bash
# texture-addresser / vector-L1 counters of the intersect kernel (one pass per group):  pmc_ta.sh <tag> <config> <frames-per-step> [bench flags...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; CFG=$2; FPS=$3; shift; shift; shift
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
ARGS="--config $CFG --steps 1 --warmup 0 --frames-per-step $FPS --no-cpu-baseline --no-roofline --streams 1 $*"
i=0
for set in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max" \
           "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $set --kernel-trace -d $OUT/set$i -o p --output-format csv -- python3 $R/bench.py $ARGS > $OUT/set$i.log 2>&1 || { echo "set $i failed"; tail -5 $OUT/set$i.log; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(f"{d}/set*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_]+)", row["Kernel_Name"]); k = m.group(1) if m else "other"
        a = agg[k][row["Counter_Name"]]; a[0] += 1; a[1] += float(row["Counter_Value"])
for k in ("k_extend_persist", "k_shade"):
    print(k)
    for c, (n, v) in sorted(agg[k].items()):
        print(f"   {c:40s} launches {n:5d}  total {v:.4g}  per launch {v / max(n, 1):.4g}")
PY
