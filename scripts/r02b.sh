#!/bin/bash
# round-2 second call: full GPU suite (multi-GPU context, bench launch forms), then A/B of intersect-kernel variants (build/ab/*.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt
one() { # lib flags...
  lib=$1; shift
  PT_HIP_LIB=$R/build/ab/$lib.so timeout -k 10 200 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$lib $* ->', d['value'], 'Ms/s  ms/step', d['ms_per_step'], ' extend avg/med', r['avg_launch_ms'], r['median_launch_ms'], ' shade avg', r['shade']['avg_launch_ms'])
"
}
for lib in m v1 v2 v3; do
  PT_HIP_LIB=$R/build/ab/$lib.so timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -k "render_parity_small or random_scenes or persistent_variants or n2_direct" > $O/parity_$lib.txt 2>&1; echo "parity $lib rc=$? $(tail -1 $O/parity_$lib.txt)"
done
for round in 1 2; do
  for lib in base m v1 v2 v3; do one $lib; done
done | tee $O/ab.txt
for cfg in C4 C5; do for lib in base v2 v3; do one $lib --config $cfg --frames-per-step 16; done; done | tee $O/ab_c45.txt
{ one v2 --refill-min 16; one v2 --refill-min 32; one v2 --path-slots 16777216; one v3 --path-slots 16777216; one v2 --none-min 16; } | tee $O/ab_opts.txt
cd /tmp && export TMPDIR=/tmp
for lib in base v2 v3; do
  PT_HIP_LIB=$R/build/ab/$lib.so rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $O/pmc_$lib -- python3 $R/bench.py --steps 1 --warmup 0 --frames-per-step 8 --no-cpu-baseline --no-roofline > $O/pmc_$lib.json 2> $O/pmc_$lib.err
  python3 - <<PY
import csv, glob, collections, re
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$O/pmc_$lib/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m=re.search(r"(k_[a-z_]+)", row["Kernel_Name"]); k=m.group(1) if m else "other"
        agg[k][row["Counter_Name"]]+=float(row["Counter_Value"])
seg=1920*1080*8*8*2*3.925
for k in ("k_extend_persist","k_shade"):
    a=agg[k]
    if a: print("$lib", k, "VALU/seg %.2f SALU/seg %.2f LDS/seg %.2f lane_util %.3f wait %.3f stall %.3f" % (a["SQ_INSTS_VALU"]/seg, a["SQ_INSTS_SALU"]/seg, a["SQ_INSTS_LDS"]/seg, a["SQ_THREAD_CYCLES_VALU"]/64/max(a["SQ_ACTIVE_INST_VALU"],1), a["SQ_WAIT_ANY"]/max(a["SQ_WAVE_CYCLES"],1), a["SQ_WAIT_INST_ANY"]/max(a["SQ_WAVE_CYCLES"],1)))
PY
  rm -rf $O/pmc_$lib
done | tee $O/pmc_ab.txt
