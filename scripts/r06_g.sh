#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06g; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for m in async1; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$m -- python3 $R/scripts/frame_mode.py $m 2 256 > $O/$m.txt 2>&1
  grep "ms/frame" $O/$m.txt
  f=$(find $O/$m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -d, -f1-7 "$f" | cut -c1-200 > $O/${m}_kernel_stats.csv
  t=$(find $O/$m -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python3 - "$t" > $O/${m}_timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ext = [r for r in rows if "pt_extend_asm" in r["Kernel_Name"]]
ext.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(ext[0]["Start_Timestamp"])
# duration and grid of the intersect launches over time, 40 samples
step = max(1, len(ext) // 60)
for r in ext[::step]:
    print(f"t {1e-6 * (int(r['Start_Timestamp']) - t0):8.1f} ms  dur {1e-3 * (int(r['End_Timestamp']) - int(r['Start_Timestamp'])):8.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size'))} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size'))} queue {r.get('Queue_Id')}")
# GPU idle: gaps on the union of all kernels
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
busy = 0; cur_s, cur_e = iv[0]
for a, b in iv[1:]:
    if a > cur_e: busy += cur_e - cur_s; cur_s, cur_e = a, b
    else: cur_e = max(cur_e, b)
busy += cur_e - cur_s
print(f"wall {1e-6 * (iv[-1][1] - iv[0][0]):.1f} ms  some kernel running {1e-6 * busy:.1f} ms")
PY
  rm -rf $O/$m
done
