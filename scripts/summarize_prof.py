#!/usr/bin/env python3
"""Condenses a scripts/profile.sh output directory into one text summary for profiles/.

usage: summarize_prof.py gpurun_out/<dir> > profiles/<name>.txt
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE counts 128-B
requests as 64 B for wide (16 B/lane) coalesced streams (MI355X_MICROARCH.md §HBM), so the summary
prints the raw figure and the x2-corrected total; k_accumulate / k_generate have known byte counts
and serve as the calibration of that correction for this code's 16-B/lane access pattern.
"""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    m = re.search(r"(k_[a-z_]+(<[^>]*>)?)", name)
    return m.group(1) if m else name.split("(")[0][-60:]


def main(d):
    print(f"# rocprofv3 summary of {d}")
    for f in glob.glob(f"{d}/bench_trace.json"):
        lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
        if lines:
            j = json.loads(lines[-1])
            print("\n## bench.py line of the traced run (rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline)")
            print(json.dumps({k: j[k] for k in ("metric", "value", "unit", "steps", "ms_per_step")}))
            print("config:", json.dumps(j.get("config")))
            print("roofline:", json.dumps(j.get("roofline")))
    for f in glob.glob(f"{d}/trace/*/*_kernel_stats.csv"):
        print("\n## kernel stats (--kernel-trace --stats)\nname,calls,total_ms,avg_us,percent,min_us,max_us")
        for row in csv.DictReader(open(f)):
            print(f"{short(row['Name'])},{row['Calls']},{float(row['TotalDurationNs'])/1e6:.3f},{float(row['AverageNs'])/1e3:.2f},{float(row['Percentage']):.3f},"
                  f"{float(row['MinNs'])/1e3:.2f},{float(row['MaxNs'])/1e3:.2f}")
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for cname, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        for f in glob.glob(f"{d}/{sub}/*/*_counter_collection.csv"):
            for row in csv.DictReader(open(f)):
                a = agg[short(row["Kernel_Name"])][row["Counter_Name"]]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    if agg:
        print("\n## HBM traffic counters (separate --pmc passes: bench.py --steps 1 --warmup 0 --frames-per-step 16 --no-roofline)")
        print("kernel,launches,FETCH_SIZE_KiB_per_launch(raw),WRITE_SIZE_KiB_per_launch,HBM_MB_per_launch(fetch x2 + write)")
        for k, v in sorted(agg.items(), key=lambda kv: -(kv[1]["FETCH_SIZE"][1] + kv[1]["WRITE_SIZE"][1])):
            if not k.startswith("k_"):
                continue
            fl, fs = v["FETCH_SIZE"]
            wl, ws = v["WRITE_SIZE"]
            n = max(fl, wl, 1)
            print(f"{k},{n},{fs/max(fl,1):.1f},{ws/max(wl,1):.1f},{(2*fs/max(fl,1)+ws/max(wl,1))*1024/1e6:.2f}")
        # per-segment HBM traffic of the wavefront kernels (PMC runs: C3, 16 frames x 8 spp, S = 3.925 segments/sample)
        seg = 1920 * 1080 * 128 * 3.925
        out = {}
        for k, v in agg.items():
            base = k.split("<")[0]
            if base in ("k_extend_persist", "k_extend", "k_shade"):
                o = out.setdefault(base, {"fetch_kib_raw": 0.0, "write_kib": 0.0})
                o["fetch_kib_raw"] += v["FETCH_SIZE"][1]
                o["write_kib"] += v["WRITE_SIZE"][1]
        for base, o in out.items():
            o["hbm_bytes_per_segment"] = round((2 * o["fetch_kib_raw"] + o["write_kib"]) * 1024 / seg, 2)
            print(f"per segment: {base} HBM bytes (FETCH x2 + WRITE) = {o['hbm_bytes_per_segment']}")
        if len(sys.argv) > 2:
            json.dump({"source": d, "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B); C3, 16 frames x 8 spp",
                       "kernels": out}, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
