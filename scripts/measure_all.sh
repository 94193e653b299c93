#!/bin/bash
# measurement pass (usage: measure_all.sh <out-tag> [pmc|rest|all]): counter summaries (final kernels) | kernel-trace stats, the bench lines, shard rehearsals, the drop-in call
# pattern, the C host, fall-back paths, fuzz.  Two gpurun calls of at most 20 minutes each: `pmc` first (the bench lines of `rest` read profiles/pmc_*.json of THIS tree).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-measure}
PART=${2:-all}
mkdir -p $O
cd $R
if [ $PART != rest ]; then
# 1. counters per configuration -> profiles/pmc_<cfg>.json (bench.py's roofline block reads them)
for c in ${PMC_CFGS:-C3 C4 C5 C2 C6}; do
  case $c in C3) f=32;; C4) f=16;; C2) f=8;; *) f=4;; esac
  FPS=$f bash scripts/pmc_all.sh ${1:-measure}/pmc_$c $c > $O/pmc_$c.log 2>&1; cp $O/pmc_$c/summary.json profiles/pmc_$c.json; echo "pmc $c done"
done
fi
[ $PART = pmc ] && exit 0
# 2. the driver's command
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_c3.json 2> $O/bench_default_c3.err; echo "bench rc=$?"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_c3_untimed_kernels.json 2>/dev/null
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams 1 > $O/bench_c3_one_stream.json 2>/dev/null
# the other configurations, each with its parity block (HIP path vs the oracle on the frames the CPU baseline renders) and cpu_baseline: the 8-GPU configs C4 / C5 on ONE GPU,
# C6 (64 BVHs + ellipsoids: the reference author's scene shape), C1 (BASELINE configs[0])
for cfg in C2 C4 C5 C6 C1; do timeout -k 10 600 python3 bench.py --config $cfg --steps 3 --warmup 1 > $O/bench_$cfg.json 2> $O/bench_$cfg.err; echo "bench $cfg rc=$?"; done
# 3. kernel trace + stats of the bench command (scripts/trace_stats.sh)
bash scripts/trace_stats.sh ${1:-measure}
# 4. tile-shard rehearsals (one shard of N alone on this GPU) and the multi-GPU context with two shards on this GPU
for n in 1 2 4 8; do timeout -k 10 300 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --rehearse-shard 0 $n 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('two streams: GPU 0 of $n ->', d['value'], 'Ms/s  ms/step', d['ms_per_step'])
"; done | tee $O/rehearsals.txt
for n in 1 2 4 8; do timeout -k 10 300 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --streams 1 --rehearse-shard 0 $n 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('one stream: shard 0 of $n ->', d['value'], 'Ms/s  ms/step', d['ms_per_step'])
"; done | tee -a $O/rehearsals.txt
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 4 --warmup 1 > $O/bench_torchrun_n1.json 2> $O/bench_torchrun_n1.err; echo "torchrun rc=$?"
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); r=d.get("roofline",{})
            print(f.split("/")[-1], d["value"], "Ms/s", d["ms_per_step"], "ms/step", "frac", r.get("frac"), "hbm", (r.get("hbm") or {}).get("frac"), "shade", (r.get("shade") or {}).get("frac"), "parity", d.get("parity"))
PY
# 5. the drop-in call pattern (one draw per frame) and the C host on ROCm's own runtime beside the Python bench
export GPU_MAX_HW_QUEUES=8
for n in 64 256; do echo "== streams 2 FRAMES $n"; FRAMES=$n timeout -k 10 300 python3 scripts/frame_loop.py 2 0 2>&1 | grep -v amdgpu.ids; done | tee $O/frame_loop.txt
echo "== streams 1 FRAMES 64" | tee -a $O/frame_loop.txt; FRAMES=64 timeout -k 10 300 python3 scripts/frame_loop.py 1 0 2>&1 | grep -v amdgpu.ids | tee -a $O/frame_loop.txt
for k in 1 2; do
  timeout -k 10 600 python3 scripts/c_host_bench.py --steps 20 --warmup 5 --streams 2 2>&1 | grep "^{" | tee -a $O/c_host.txt
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(json.dumps({'host': 'python bench.py', 'hip_runtime': d['hip_runtime'], 'ms_per_step': d['ms_per_step'], 'value': d['value']}))" | tee -a $O/c_host.txt
done
# 6. the launches the hand-written kernel does not take, and a fuzz campaign over the switches
timeout -k 10 400 python3 scripts/fallback_paths.py 2>&1 | grep -v amdgpu.ids | tee $O/fallback_paths.txt
timeout -k 10 600 python3 scripts/fuzz_campaign.py 70001 400 2>&1 | tail -3 | tee $O/fuzz.txt
