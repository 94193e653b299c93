#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command -> gpurun_out/<tag>/trace_stats.txt (per-kernel calls / average duration + the bench line's own
# HIP-event figures: the average pt_extend_asm duration must agree with roofline.avg_launch_ms).  --no-alone-pass: without it the trace also
# holds the shorter launches of the one-stream pass that follows the timed region.      usage: scripts/trace_stats.sh <out-tag> [bench flags]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-trace}; shift
mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-alone-pass "$@" > $O/bench_trace.json 2> $O/bench_trace.err )
python3 - $O <<'PY' > $O/trace_stats.txt
import csv, glob, json, sys
O = sys.argv[1]
print("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-alone-pass")
for f in glob.glob(O + "/trace/**/*kernel_stats.csv", recursive=True):
    print("name,calls,total_ms,avg_us,percent,min_us,max_us")
    for row in csv.DictReader(open(f)):
        n = row["Name"]; i = n.find("k_") if "pt_extend_asm" not in n else n.find("pt_extend_asm"); n = n[i:i + 40] if i >= 0 else n[:40]
        print("%s,%s,%.3f,%.2f,%.3f,%.2f,%.2f" % (n, row["Calls"], float(row["TotalDurationNs"]) / 1e6, float(row["AverageNs"]) / 1e3, float(row["Percentage"]),
                                                   float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3))
for l in open(O + "/bench_trace.json"):
    if l.startswith("{"):
        d = json.loads(l)
        print("bench line of the traced run:", json.dumps({k: d[k] for k in ("value", "ms_per_step", "steps")}), json.dumps({k: d["roofline"][k] for k in ("avg_launch_ms", "launches", "frac")}),
              "shade avg", d["roofline"]["shade"]["avg_launch_ms"])
PY
rm -rf $O/trace
cat $O/trace_stats.txt
