#!/bin/bash
# round-2 opening measurement: sanity tests, baseline bench lines, pool-size sweep, counter passes for C3/C4/C5
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02a
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.txt
timeout -k 10 300 python3 bench.py --steps 8 --warmup 2 > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc=$?"
for ps in 16777216 33554432; do
  timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --path-slots $ps > $O/bench_c3_pool$ps.json 2> $O/bench_c3_pool$ps.err; echo "pool $ps rc=$?"
done
for cfg in C2 C4 C5; do
  timeout -k 10 400 python3 bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_$cfg.json 2> $O/bench_$cfg.err; echo "bench $cfg rc=$?"
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); r=d.get("roofline",{})
            print(f.split("/")[-1], d["value"], "Ms/s", d["ms_per_step"], "ms/step ext", r.get("avg_launch_ms"), r.get("launches"), "shade", r.get("shade_avg_launch_ms"), "S", r.get("segments_per_sample"))
PY
for cfg in C3 C4 C5; do
  bash $R/scripts/pmc_all.sh r02a/pmc_$cfg $cfg > $O/pmc_$cfg.log 2>&1; echo "pmc $cfg done"
done
