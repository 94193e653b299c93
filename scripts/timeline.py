#!/usr/bin/env python3
"""Per-iteration timeline of one wavefront batch from a rocprofv3 --kernel-trace CSV.

usage: timeline.py <dir with *_kernel_trace.csv> [window]      window 0: whole trace; k > 0: after the k-th k_accumulate;
                                                               k < 0: from the k_accumulate before the k-th last one
Prints, for the chosen window, one line per group of 8 iterations: the extend and shade
durations, the idle gap between kernels, and the running clock - shows where a batch spends its time (ramp, steady state, drain).
"""
import csv
import glob
import re
import sys


def main(d, which=0, group=8):
    f = glob.glob(f"{d}/**/*_kernel_trace.csv", recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_]+)", r["Kernel_Name"])
        if m:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1)))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if r[2] == "k_frame_setup"]
    # a batch = from a k_generate launch to the k_accumulate that follows
    accs = [i for i, r in enumerate(rows) if r[2] == "k_accumulate"]
    # window: from the accumulate launch number `which` (negative: from the end) to the last launch of the trace
    batch = rows if which == 0 else (rows[accs[which] + 1:] if which > 0 else rows[(accs[which - 1] + 1 if len(accs) >= -which + 1 else 0):])
    t0 = batch[0][0]
    print(f"batch: {len(batch)} launches, span {(batch[-1][1] - t0) / 1e6:.3f} ms, busy {sum(e - s for s, e, _ in batch) / 1e6:.3f} ms")
    by = {}
    for s, e, n in batch:
        by.setdefault(n, [0, 0]); by[n][0] += 1; by[n][1] += e - s
    for n, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print(f"  {n:20s} {c:5d} launches {t / 1e6:8.3f} ms  avg {t / c / 1e3:8.2f} us")
    it, acc, prev_end = 0, None, batch[0][1]
    print("iter   clock_ms  extend_us  shade_us  other_us  gap_us   (per iteration, averaged over groups of %d)" % group)
    ext = sh = oth = gap = 0.0
    n_in = 0
    for s, e, n in batch[1:]:
        gap += max(0, s - prev_end); prev_end = max(prev_end, e)
        if n.startswith("k_extend"):
            ext += e - s
        elif n == "k_shade":
            sh += e - s; it += 1; n_in += 1
            if n_in == group:
                print(f"{it:5d} {(e - t0) / 1e6:9.3f} {ext / group / 1e3:9.1f} {sh / group / 1e3:9.1f} {oth / group / 1e3:9.1f} {gap / group / 1e3:7.1f}")
                ext = sh = oth = gap = 0.0; n_in = 0
        else:
            oth += e - s
            if n in ("k_accumulate", "k_revive", "k_submit"):
                print(f"      {(s - t0) / 1e6:9.3f} ms  {n}  ({(e - s) / 1e3:.1f} us)")
    if n_in:
        print(f"{it:5d} {(prev_end - t0) / 1e6:9.3f} {ext / n_in / 1e3:9.1f} {sh / n_in / 1e3:9.1f} {oth / n_in / 1e3:9.1f} {gap / n_in / 1e3:7.1f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
