cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/${1:-trace_other}; mkdir -p $O
for cfg in C4 C5 C6 C2; do
  fps=16; [ $cfg = C5 ] && fps=4; [ $cfg = C2 ] && fps=8; [ $cfg = C6 ] && fps=4
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$cfg -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --frames-per-step $fps --steps 3 --warmup 1 --no-cpu-baseline --no-alone-pass > $O/bench_trace_$cfg.json 2> $O/bench_trace_$cfg.err )
  python3 - <<PY
import csv, glob, json
print("## $cfg: rocprofv3 --kernel-trace --stats -- python3 bench.py --config $cfg --frames-per-step $fps --steps 3 --warmup 1 --no-cpu-baseline --no-alone-pass")
for f in glob.glob("$O/trace_$cfg/**/*kernel_stats.csv", recursive=True):
    print("name,calls,total_ms,avg_us,percent,min_us,max_us")
    for row in csv.DictReader(open(f)):
        n=row["Name"]; i=n.find("k_") if "pt_extend_asm" not in n else n.find("pt_extend_asm"); n=n[i:i+44] if i>=0 else n[:44]
        print(f"{n},{row['Calls']},{float(row['TotalDurationNs'])/1e6:.3f},{float(row['AverageNs'])/1e3:.2f},{float(row['Percentage']):.3f},{float(row['MinNs'])/1e3:.2f},{float(row['MaxNs'])/1e3:.2f}")
for l in open("$O/bench_trace_$cfg.json"):
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]; print("bench line of the traced run:", json.dumps({k:d[k] for k in ("value","ms_per_step","steps")}), json.dumps({k:r[k] for k in ("avg_launch_ms","launches","frac")}), "shade avg", r["shade"]["avg_launch_ms"], "chip", json.dumps({"valu": (r.get("chip_valu_issue") or {}).get("frac"), "hbm": r["hbm"]["frac"]}))
PY
  rm -rf $O/trace_$cfg
done
