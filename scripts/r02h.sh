#!/bin/bash
# round-2: full GPU suite with the multi-stream default, smoke, the driver's bench command, stream-count check per config
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02h
mkdir -p $O
cd $R
timeout -k 10 1100 python3 -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.txt
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_c3.json 2> $O/bench_default_c3.err; echo "bench rc=$?"
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --streams 1 > $O/bench_c3_one_stream.json 2> $O/bench_c3_one_stream.err; echo "bench 1 stream rc=$?"
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 8 --warmup 2 > $O/bench_torchrun_n1.json 2> $O/bench_torchrun_n1.err; echo "torchrun rc=$?"
for cfg in C2 C4 C5; do timeout -k 10 600 python3 bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_$cfg.json 2> $O/bench_$cfg.err; echo "bench $cfg rc=$?"; done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); r=d.get("roofline",{}); c=r.get("chip",{})
            print(f.split("/")[-1], d["value"], "Ms/s", d["ms_per_step"], "ms/step", "frac", r.get("frac"), "chip valu", (c.get("valu_issue") or {}).get("frac"), "chip hbm", (c.get("hbm") or {}).get("frac"), "conc", c.get("kernel_concurrency"), "parity", d.get("parity"))
PY
