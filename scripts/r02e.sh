#!/bin/bash
# round-2 fifth call: full GPU suite on the default build, then default vs deferred hit-record stores (speed, WRITE_SIZE)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02e
mkdir -p $O
cd $R
timeout -k 10 1000 python3 -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt
one() { # lib flags...
  lib=$1; shift
  PT_HIP_LIB=$R/build/ab/$lib.so timeout -k 10 300 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$lib $* ->', d['value'], 'Ms/s  ms/step', d['ms_per_step'], ' extend avg/med', r['avg_launch_ms'], r['median_launch_ms'], ' shade avg', r['shade']['avg_launch_ms'])
"
}
PT_HIP_LIB=$R/build/ab/E1.so timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -x -q > $O/parity_E1.txt 2>&1; echo "parity E1 rc=$? $(tail -1 $O/parity_E1.txt)"
for round in 1 2 3; do for lib in E0 E1; do one $lib; done; done | tee $O/ab.txt
for cfg in C4 C5 C2; do for lib in E0 E1; do one $lib --config $cfg --frames-per-step 16; done; done | tee $O/ab_c45.txt
cd /tmp && export TMPDIR=/tmp
for lib in E0 E1; do
  for ctr in WRITE_SIZE FETCH_SIZE; do
  PT_HIP_LIB=$R/build/ab/$lib.so rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_$lib$ctr -- python3 $R/bench.py --steps 1 --warmup 0 --frames-per-step 8 --no-cpu-baseline --no-roofline > $O/pmc_$lib$ctr.json 2> $O/pmc_$lib$ctr.err
  python3 - <<PY
import csv, glob, collections, re
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$O/pmc_$lib$ctr/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m=re.search(r"(k_[a-z_]+)", row["Kernel_Name"]); k=m.group(1) if m else "other"
        agg[k][row["Counter_Name"]]+=float(row["Counter_Value"])
seg=1920*1080*8*8*2*3.925
for k in ("k_extend_persist","k_shade"):
    for c,v in agg[k].items(): print("$lib", k, c, "%.2f B/segment" % (v*1024/seg))
PY
  rm -rf $O/pmc_$lib$ctr
  done
done | tee $O/pmc_ab.txt
