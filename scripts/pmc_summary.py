#!/usr/bin/env python3
"""Condenses the counter passes of scripts/pmc_all.sh into summary.json (+ a text table on stdout).

usage: pmc_summary.py gpurun_out/<tag> <config> <frames-per-step>

Per kernel: counter totals over all launches of a pass, and the figures bench.py's `roofline` block is built from:
  valu_per_segment   SQ_INSTS_VALU / segments                (wave-instructions; a v_pk_* counts once)
  salu_per_segment   SQ_INSTS_SALU / segments
  lane_util          SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)   lanes busy per issued VALU instruction
  wait_share         SQ_WAIT_ANY / SQ_WAVE_CYCLES            wave parked in s_waitcnt / barrier
  issue_stall_share  SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES       wave ready, issue port taken
  active_share       SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  hbm_bytes_per_segment   (2 * FETCH_SIZE + WRITE_SIZE) KiB * 1024 / segments   (gfx950: FETCH_SIZE tallies 128-B requests as 64 B,
                           /opt/skills/guides/MI355X_MICROARCH.md, HBM)
  (round 6, the CU's vector-memory path; *_sum counters are sums over the 256 CUs' units, GRBM_GUI_ACTIVE over the 8 XCDs; rocprofv3 serialises the
   kernels of a counter pass, so these are each kernel ALONE on the chip)
  ta_busy_cycles_per_segment / td_busy_cycles_per_segment   TA_TA_BUSY_sum / segments, TD_TD_BUSY_sum / segments
  ta_busy_frac_alone / td_busy_frac_alone                    the same over 256 x (GRBM_GUI_ACTIVE / 8): the share of the kernel's cycles its CU's unit is busy
  tag_lookups_per_vmem_inst   TCP_TOTAL_CACHE_ACCESSES_sum / (SQ_INSTS_VMEM_RD + SQ_INSTS_VMEM_WR)     (a coalesced 16-B-per-lane load: 4-8)
  l1_hit_rate                 1 - TCP_TCC_READ_REQ_sum / TCP_TOTAL_CACHE_ACCESSES_sum;   l2_round_trip_cycles   TCP_TCC_READ_REQ_LATENCY_sum / TCP_TCC_READ_REQ_sum
segments = W * H * SAMPLE_RES * frames-per-step * 2 passes (bench.py's untimed set-up pass + 1 timed step) * S, with S = segments per
sample from the statistics pass of the un-profiled run of the same command (plain.json; deterministic per scene and seeds).
VALU issue roof: 256 CUs * 4 SIMDs * 2.4 GHz / 2 cycles per wave64 instruction = 1.2288e12 wave-instructions/s.
"""
import collections
import csv
import glob
import importlib.util
import json
import os
import re
import sys


def kernel_source_hash():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("pt_build", os.path.join(root, "pathtracer-0_amd", "build.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m.kernel_source_hash()

VALU_PEAK = 256 * 4 * 2.4e9 / 2
HBM_PEAK = 8.0e12
UNIT_PEAK = 256 * 2.4e9           # busy cycles per second of a per-CU unit (TA, TD) summed over the chip, at the peak engine clock


def main(d, cfg, fps):
    plain = None
    for l in open(f"{d}/plain.json").read().splitlines():
        if l.startswith("{"):
            plain = json.loads(l)
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(f"{d}/set*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(r"(k_[a-z_]+|pt_extend_asm)", row["Kernel_Name"])
            k = m.group(1) if m else "other"
            a = agg[k][row["Counter_Name"]]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    out = {"source": d, "config": cfg, "frames_per_step": fps, "command": f"rocprofv3 --pmc <set> -- python3 bench.py --config {cfg} --steps 1 --warmup 0 "
           f"--frames-per-step {fps} --no-cpu-baseline --no-roofline", "valu_issue_peak_per_s": VALU_PEAK,
           "kernel_source_hash": kernel_source_hash(), "kernels": {}}
    if plain:
        rf = plain.get("roofline", {})
        W, H = plain["config"]["width"], plain["config"]["height"]
        S = rf.get("segments_per_sample")
        sres = plain["config"]["spp_per_step"] // fps
        seg = W * H * sres * fps * 2 * S
        out.update({"segments_per_sample": S, "segments_in_a_pass": seg, "plain_run": {"value": plain["value"], "ms_per_step": plain["ms_per_step"],
                    "extend_avg_launch_ms": (rf.get("in_run") or rf).get("avg_launch_ms"), "shade_avg_launch_ms": ((rf.get("shade") or {}).get("in_run") or rf.get("shade") or {}).get("avg_launch_ms"),
                    "extend_launches": (rf.get("in_run") or rf).get("launches"),
                    "per_segment": rf.get("per_segment")}})
    else:
        seg = None
    for k, v in sorted(agg.items()):
        if k not in ("pt_extend_asm", "k_extend_persist", "k_extend", "k_shade", "k_accumulate", "k_revive"):
            continue
        tot = {c: x[1] for c, x in v.items()}
        n = max(x[0] for x in v.values())
        o = {"launches_per_pass": n, "totals": {c: round(x) for c, x in sorted(tot.items())}}
        g = tot.get
        if seg and k in ("pt_extend_asm", "k_extend_persist", "k_extend", "k_shade"):
            if g("SQ_INSTS_VALU"):
                o["valu_per_segment"] = round(g("SQ_INSTS_VALU") / seg, 3)
                o["salu_per_segment"] = round(g("SQ_INSTS_SALU", 0) / seg, 3)
                o["lds_insts_per_segment"] = round(g("SQ_INSTS_LDS", 0) / seg, 3)
            if g("SQ_INSTS_VMEM_RD") is not None and g("SQ_INSTS_VMEM_RD"):
                o["vmem_rd_per_segment"] = round(g("SQ_INSTS_VMEM_RD") / seg, 3)
                o["vmem_wr_per_segment"] = round(g("SQ_INSTS_VMEM_WR", 0) / seg, 3)
            if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
                o["hbm_fetch_bytes_per_segment_x2"] = round(2 * g("FETCH_SIZE") * 1024 / seg, 2)
                o["hbm_write_bytes_per_segment"] = round(g("WRITE_SIZE") * 1024 / seg, 2)
                o["hbm_bytes_per_segment"] = round((2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024 / seg, 2)
            if g("TA_TA_BUSY_sum"):
                o["ta_busy_cycles_per_segment"] = round(g("TA_TA_BUSY_sum") / seg, 3)
            if g("TD_TD_BUSY_sum"):
                o["td_busy_cycles_per_segment"] = round(g("TD_TD_BUSY_sum") / seg, 3)
        if g("GRBM_GUI_ACTIVE"):
            unit_cycles = 256.0 * g("GRBM_GUI_ACTIVE") / 8.0
            if g("TA_TA_BUSY_sum"):
                o["ta_busy_frac_alone"] = round(g("TA_TA_BUSY_sum") / unit_cycles, 4)
            if g("TD_TD_BUSY_sum"):
                o["td_busy_frac_alone"] = round(g("TD_TD_BUSY_sum") / unit_cycles, 4)
                o["td_waiting_for_cache_frac_alone"] = round(g("TD_TC_STALL_sum", 0) / unit_cycles, 4)
            if g("TA_ADDR_STALLED_BY_TC_CYCLES_sum") is not None:
                o["ta_addr_stalled_by_cache_frac_alone"] = round(g("TA_ADDR_STALLED_BY_TC_CYCLES_sum", 0) / unit_cycles, 4)
        if g("TCP_TOTAL_CACHE_ACCESSES_sum"):
            vm = g("SQ_INSTS_VMEM_RD", 0) + g("SQ_INSTS_VMEM_WR", 0)
            if vm:
                o["tag_lookups_per_vmem_inst"] = round(g("TCP_TOTAL_CACHE_ACCESSES_sum") / vm, 2)
            o["l1_hit_rate"] = round(1.0 - g("TCP_TCC_READ_REQ_sum", 0) / g("TCP_TOTAL_CACHE_ACCESSES_sum"), 4)
            if g("TCP_TCC_READ_REQ_sum"):
                o["l2_round_trip_cycles"] = round(g("TCP_TCC_READ_REQ_LATENCY_sum", 0) / g("TCP_TCC_READ_REQ_sum"), 1)
        if g("SQ_ACTIVE_INST_VALU"):
            o["lane_util"] = round(g("SQ_THREAD_CYCLES_VALU", 0) / (64.0 * g("SQ_ACTIVE_INST_VALU")), 4)
        if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_ANY") is not None:
            pass
        out["kernels"][k] = o
    # shares need SQ_WAVE_CYCLES from set 1 and the wait counters from set 2: both are totals of the same launches
    for k, o in out["kernels"].items():
        t = o["totals"]
        wc = t.get("SQ_WAVE_CYCLES")
        if wc:
            for name, c in (("wait_share", "SQ_WAIT_ANY"), ("issue_stall_share", "SQ_WAIT_INST_ANY"), ("active_share", "SQ_ACTIVE_INST_ANY"),
                            ("valu_active_share", "SQ_ACTIVE_INST_VALU"), ("scalar_active_share", "SQ_ACTIVE_INST_SCA")):
                if c in t:
                    o[name] = round(t[c] / wc, 4)
    ext = "pt_extend_asm" if "pt_extend_asm" in out["kernels"] else "k_extend_persist"      # the hand-written intersect kernel, or the compiled one
    if plain and ext in out["kernels"]:
        o = out["kernels"][ext]
        pr = out["plain_run"]
        if pr["extend_avg_launch_ms"] and pr["extend_launches"] and "valu_per_segment" in o:
            seg_rate = (seg / 2.0) / (pr["extend_avg_launch_ms"] * 1e-3 * pr["extend_launches"])      # segments/s while the kernel runs (timed step of the plain run)
            o["segments_per_s_in_kernel"] = seg_rate
            o["valu_issue_frac"] = round(o["valu_per_segment"] * seg_rate / VALU_PEAK, 4)
            for u in ("ta", "td"):                   # the unit's busy cycles over 256 units x 2.4 GHz while the kernel runs beside the other stream (the plain run's launch durations)
                if f"{u}_busy_cycles_per_segment" in o:
                    o[f"{u}_busy_frac_in_run"] = round(o[f"{u}_busy_cycles_per_segment"] * seg_rate / UNIT_PEAK, 4)
            if "hbm_bytes_per_segment" in o:
                o["hbm_frac"] = round(o["hbm_bytes_per_segment"] * seg_rate / HBM_PEAK, 4)
    json.dump(out, open(f"{d}/summary.json", "w"), indent=1)
    print(f"# counter summary {cfg} ({d}); segments per pass = {seg}")
    for k, o in out["kernels"].items():
        print(k, json.dumps({a: b for a, b in o.items() if a != "totals"}))
        print("   totals", json.dumps(o["totals"]))
    if plain:
        print("plain run:", json.dumps(out["plain_run"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]))
