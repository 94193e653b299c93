#!/bin/bash
# SQ counter passes on a short bench run (each --pmc set in its own run; no trace flags).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc_sq}
shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_VMEM_WR SQ_WAVES_EQ_64 SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/set$i -- python3 $R/bench.py --steps 1 --warmup 0 --frames-per-step 1 --no-cpu-baseline --no-roofline "$@" > $OUT/set$i.json 2> $OUT/set$i.err || echo "set $i failed"
done
python3 - <<PY
import csv, glob, collections, re
agg=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
for f in glob.glob("$OUT/set*/*/*_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        m=re.search(r"(k_[a-z_]+)", row["Kernel_Name"]); k=m.group(1) if m else "other"
        a=agg[k][row["Counter_Name"]]; a[0]+=1; a[1]+=float(row["Counter_Value"])
for k in ("k_extend","k_extend_persist","k_shade"):
    if k in agg:
        print(k, {c: round(v[1]/v[0],1) for c,v in sorted(agg[k].items())}, "launches", max(v[0] for v in agg[k].values()))
PY
