#!/usr/bin/env python3
"""How busy is the GPU, and with what, in a rocprofv3 --kernel-trace of a bench run with several streams?

usage: overlap.py <dir with *_kernel_trace.csv>
Over the steady part of the trace (between the second and the last k_accumulate): the time with 0 / 1 / 2+ kernels in flight,
split by which kinds run together, and per stream (HIP queue) the gap between the end of one kernel and the start of the next.
"""
import collections
import csv
import glob
import re
import sys

d = sys.argv[1]
f = glob.glob(f"{d}/**/*_kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    m = re.search(r"(k_[a-z_]+)", r["Kernel_Name"])
    if m:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1), r.get("Queue_Id", "0")))
rows.sort()
accs = [r[1] for r in rows if r[2] == "k_accumulate"]
t0, t1 = accs[1], accs[-1]
rows = [r for r in rows if r[0] >= t0 and r[1] <= t1]
ev = []
for s, e, n, q in rows:
    k = "E" if n.startswith("k_extend") else "S" if n == "k_shade" else "o"
    ev.append((s, 1, k)); ev.append((e, -1, k))
ev.sort()
cur = collections.Counter(); last = t0; acc = collections.Counter()
for t, dlt, k in ev:
    key = "".join(sorted(k2 * c for k2, c in cur.items())) or "idle"
    acc[key] += t - last; last = t
    cur[k] += dlt
    if cur[k] == 0: del cur[k]
tot = t1 - t0
print(f"window {tot / 1e6:.1f} ms, {len(rows)} launches")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    if v / tot > 0.002: print(f"  in flight {k:8s} {100.0 * v / tot:5.1f} %")
byq = collections.defaultdict(list)
for s, e, n, q in rows: byq[q].append((s, e, n))
for q, lst in byq.items():
    gaps = [lst[i + 1][0] - max(x[1] for x in lst[:i + 1][-3:]) for i in range(len(lst) - 1)]
    busy = sum(e - s for s, e, _ in lst)
    g = sorted(gaps)
    print(f"  queue {q}: {len(lst)} launches, busy {100.0 * busy / tot:.1f} % of the window, gap between kernels: median {g[len(g) // 2] / 1e3:.1f} us, mean {sum(max(x, 0) for x in g) / len(g) / 1e3:.1f} us, "
          f"p90 {g[int(len(g) * 0.9)] / 1e3:.1f} us")
