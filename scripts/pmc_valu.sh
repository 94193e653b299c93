#!/bin/bash
# one PMC pass: VALU instruction counts + lane utilisation of the intersect kernel for a bench variant
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc_valu}
shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $OUT/set1 -- python3 $R/bench.py --steps 1 --warmup 0 --frames-per-step 2 --no-cpu-baseline --no-roofline "$@" > $OUT/set1.json 2> $OUT/set1.err || echo failed
python3 - <<PY
import csv, glob, collections, re
agg=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
for f in glob.glob("$OUT/set1/*/*_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        m=re.search(r"(k_[a-z_]+)", row["Kernel_Name"]); k=m.group(1) if m else "other"
        a=agg[k][row["Counter_Name"]]; a[0]+=1; a[1]+=float(row["Counter_Value"])
SEG = 1920*1080*16*3.925          # segments of the run (2 frames x 8 spp, C3: 3.925 segments/sample)
for k in ("k_extend","k_extend_persist","k_shade"):
    if k in agg:
        d={c: v[1] for c,v in agg[k].items()}
        print("$*", k, "launches", int(max(v[0] for v in agg[k].values())), "per segment: VALU wave-instr x64 = %.0f lane-slots, useful lane-instr %.0f, SALU x64 %.0f, LDS x64 %.0f | util %.1f%% wait %.0f%%" % (
              d["SQ_INSTS_VALU"]*64/SEG, d["SQ_THREAD_CYCLES_VALU"]/SEG, d["SQ_INSTS_SALU"]*64/SEG, d["SQ_INSTS_LDS"]*64/SEG,
              100*d["SQ_THREAD_CYCLES_VALU"]/(d["SQ_ACTIVE_INST_VALU"]*64), 100*d["SQ_WAIT_ANY"]/d["SQ_WAVE_CYCLES"]))
PY
