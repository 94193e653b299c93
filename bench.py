#!/usr/bin/env python3
"""bench.py — Msamples/s of the wavefront path tracer on BASELINE.json's headline workload.

    python bench.py --gpus N --steps K --warmup W

Every GPU runs --streams (default 2) independent wavefront streams — tile shards with their own path pool and HIP stream whose
kernels overlap freely (pt_create_multi with a device listed twice): the intersect kernel of one stream is instruction-issue bound,
the shading kernel of the other HBM bound, and the launch tails fill (C3 +9 %, C4 +24 % over one stream).
N > 1 runs in either of two forms, same tile sharding, same single RCCL gather per image:
  * started plainly (`python bench.py --gpus N`): ONE process, ONE context behind the C ABI (pt_create_multi: a host thread per
    stream inside the library, ncclGather on device 0) — what the reference's single-threaded Java host would call;
  * under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` (the driver's scaling runs): one process per GPU
    (pt_create_multi_part: its streams are shards rank*K .. rank*K+K-1 of N*K), torch.distributed "nccl" (= RCCL) gather of the packed
    blocks, the library's un-tiling kernel on rank 0, barrier + max-over-ranks timing.

One step = one image of the workload: `frames_per_step` frames x SAMPLE_RES samples/pixel over the whole W x H image (tile-sharded
over the N GPUs) and the single gather of the accumulated framebuffer on GPU 0.  Consecutive steps overlap on the GPU (a step's
last paths finish underneath the next step's first ones; its image is gathered two steps later); the K steps, their K gathers and
the final drain all lie inside the timed region.  --sync: no overlap.  Default workload C3 = BASELINE.json configs[2] (1920x1080,
8 bounces, glass + metal spheres: the configuration the metric "Msamples/s at 1920x1080x8-bounce" is quoted on; fits one GPU),
32 frames x 8 spp = 256 spp per step.  Inputs (scene, path pool, accumulators) are resident in HBM before the timed region.
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The streams of a GPU must land on different hardware queues to overlap.  Left to its default the HIP runtime put both streams of a
# process that also hosts torch's and RCCL's streams on ONE queue (kernel concurrency 1.1 instead of 2.0, profiles/r02_i_*); with the
# variable set — read once, when the process initialises HIP — they are dealt round-robin.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

try:
    METRIC = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]      # "Msamples/sec (whole node) at 1920x1080x8-bounce; per-pixel RMSE vs ref"
except Exception:
    METRIC = "Msamples/sec (whole node) at 1920×1080×8-bounce; per-pixel RMSE vs ref"
HBM_PEAK_GBS = 8000.0                      # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
UNIT_PEAK_GCYC = 256 * 2.4                 # busy cycles/ns of a per-CU unit of the vector-memory path (TA, TD) summed over the chip: 256 CUs x 2.4 GHz = 614.4 G/s
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2        # wave64 VALU instructions/ns the chip can issue: 256 CUs x 4 SIMD-32 x 2.4 GHz / 2 cycles each = 1228.8 G/s
Q_EXTEND = 44                              # algorithmic queue bytes per segment in the intersect kernel: read O,D (24) + write hit record (20), SURVEY.md §8(d)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C3", choices=["C1", "C2", "C3", "C4", "C5", "C6"])
    ap.add_argument("--frames-per-step", type=int, default=None, help="frames (x SAMPLE_RES spp) per step; default = the config's full spp")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--path-slots", type=int, default=None)
    ap.add_argument("--sync", action="store_true", help="drain the path pool at the end of every step (pt_render_batch) instead of overlapping consecutive steps")
    ap.add_argument("--max-batch", type=int, default=32, help="frames per wavefront batch (bounds the per-frame staging buffer: 16 B x pixels x frames)")
    ap.add_argument("--streams", type=int, default=2, help="independent wavefront streams per GPU (each a tile shard with its own path pool and HIP stream)")
    ap.add_argument("--devices", default=None, help="single-process multi-GPU context on these HIP devices, e.g. 0,1,2,3 (default 0..N-1); a device listed "
                                                    "twice (0,0) rehearses the sharding on one GPU (gather by device copies: RCCL refuses duplicate devices)")
    ap.add_argument("--spawn", action="store_true", help="N > 1 started plainly: instead of ONE process driving all GPUs through pt_create_multi, start "
                    "`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child (before this process touches HIP) and relay its line")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"], help="gloo: the process-per-GPU form with the collective on CPU tensors — "
                    "rehearses every line of the N > 1 path where the ranks cannot have a GPU each (with PT_BENCH_ONE_GPU=1 all ranks use GPU 0)")
    ap.add_argument("--dist", action="store_true", help="take the torch.distributed path even at WORLD_SIZE 1 (exercises the RCCL gather of the process-per-GPU form on one GPU)")
    for name in ("lds-budget", "extend-mode", "extend-tpb", "extend-cache", "refill-min", "none-min", "extend-blocks-per-cu", "inner-keep", "bfs-nodes", "stack-mode", "asm-loop", "asm-tpb", "asm-node-layout", "asm-root-cull", "cu-partition"):
        ap.add_argument("--" + name, type=int, default=None)
    ap.add_argument("--rehearse-shard", type=int, nargs=2, metavar=("RANK", "COUNT"), default=None,
                    help="single-process rehearsal of ONE tile shard of a COUNT-GPU run (no collective); reports that shard's rate")
    ap.add_argument("--contract", default="exact", choices=["exact", "fast"], help="numeric contract: exact (default; framebuffers bit-identical to the oracle) or the "
                    "opt-in relaxed one (hardware rcp/rsq/sqrt/log/cos in Box-Muller, normalize, 1/d, 1/det; per-pixel RMSE <= 1e-3, reported in the line)")
    ap.add_argument("--sky", type=int, nargs=2, metavar=("W", "H"), default=None, help="replace the workload's 1x1 sky texel by a synthetic W x H equirect image (the reference binds a "
                    "4096 x 2048 one as texture 0, dispatch.java:221) and move the camera back so that a third of the primary rays see it directly")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-kernel HIP-event timing (events add launch gaps)")
    ap.add_argument("--no-alone-pass", action="store_true", help="skip the one-stream pass after the timed region (the line then has no roofline.alone block): "
                    "for kernel traces whose per-kernel averages are to be compared with roofline.avg_launch_ms")
    return ap.parse_args()


def host_cores():
    """(threads to use, what bounds them): the host's cores as far as this process may use them — os.cpu_count(), the affinity mask and the
    cgroup CPU quota, whichever is smallest."""
    total = os.cpu_count() or 1
    info = {"cores_available": total}
    n = total
    try:
        aff = len(os.sched_getaffinity(0))
        info["affinity"] = aff
        n = min(n, aff)
    except Exception:
        pass
    quota = None
    try:                                        # cgroup v2: "<quota> <period>" or "max <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:                                    # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        info["cgroup_cpu_quota"] = round(quota, 2)
        n = min(n, max(1, int(quota + 0.5)))
    return max(1, n), info


def roofline_block(args, r, stats, samples, world, dt, value, sample_res, n_gpus=1, ext_kernel="k_extend_persist", alone=None):
    """Bounds of the dominant kernel (the intersect kernel) and of the run, each recomputable from tracked files: the per-segment counter
    figures come from the committed rocprofv3 summary profiles/pmc_<config>.json (scripts/pmc_all.sh: SQ_INSTS_VALU, SQ_THREAD_CYCLES_VALU,
    FETCH_SIZE, WRITE_SIZE ... per segment), the segments per launch and the launch durations are measured live (statistics pass + HIP events
    on the launch streams).  Every `frac` in the block is a fraction of a roof the hardware can actually reach, so it is <= 1.

    Top level = the dominant kernel in the TIMED configuration against the resource that binds it — vector-instruction issue (its node and
    triangle records come from LDS / L1 / L2):
      achieved   SQ_INSTS_VALU per segment x segments per launch / mean launch duration of the timed region (HIP events on the launch streams)
      traffic    HBM bytes per launch of that kernel from the counters (FETCH_SIZE x 2 + WRITE_SIZE, the guide's gfx950 correction)
      hbm_frac   north_star's figure: rocprofv3 HBM bytes of BOTH kernels over the wall time of the timed region, per GPU, / 8 TB/s (details under `hbm`)
      hbm        the same against the HBM roof in the bench contract's form (bound / achieved / peak / unit / frac / traffic), and per kernel
      algorithmic  SURVEY.md §8(d)'s bytes (44 B queue + 44 B per node visit + 36 B per triangle test + 124 B per hit update, the REFERENCE's
                 buffer layout streamed from memory) over the same launch durations: a bookkeeping figure — the device-private BVH is served from
                 LDS / L1 / L2, these bytes mostly never reach HBM, and for a cache-resident tree the rate exceeds the HBM peak.  No `frac`.
      alone      the kernels alone on the chip (a one-stream pass after the timed region), shade = k_shade."""
    n_ext, ms_ext = r.kernel_time("extend")
    n_sh, ms_sh = r.kernel_time("shade")
    S = stats["segments"] / max(stats["samples"], 1)
    seg = S * samples                                   # segments traced in the timed region by all devices (S from the statistics pass)
    nv, tt, hu = stats["nodes"] / stats["segments"], stats["tritests"] / stats["segments"], stats["hitupd"] / stats["segments"]
    avg_ext, avg_sh = ms_ext / max(n_ext, 1), ms_sh / max(n_sh, 1)       # mean launch over all devices' launches
    seg_per_launch = seg / max(n_ext, 1)
    seg_rate = seg_per_launch / (avg_ext * 1e-3) if avg_ext > 0 else 0.0       # segments/s per stream while the intersect kernel runs
    seg_rate_sh = (seg / max(n_sh, 1)) / (avg_sh * 1e-3) if avg_sh > 0 else 0.0
    prof, prof_name = None, f"profiles/pmc_{args.config}.json"
    if os.path.exists(os.path.join(ROOT, prof_name)):
        prof = json.load(open(os.path.join(ROOT, prof_name)))
    from pathtracer_0_amd import build as _build
    src_hash = _build.kernel_source_hash()
    stale = prof is not None and prof.get("kernel_source_hash") != src_hash      # the per-segment counter figures were measured on other kernels
    streams = max(world // max(n_gpus, 1), 1)
    if alone is None and streams == 1:
        alone = {"avg_ext_ms": avg_ext, "avg_shade_ms": avg_sh, "seg_per_launch": seg_per_launch, "launches": n_ext, "from": "the timed region (one stream per GPU)"}
    ke = (prof or {}).get("kernels", {}).get(ext_kernel)
    ks = (prof or {}).get("kernels", {}).get("k_shade")
    # algorithmic bytes per segment of the intersect kernel / per sample of the whole path, in the reference's layout
    b_ext = Q_EXTEND + nv * 44 + tt * 36 + hu * 124
    b_samp = S * 304 + S * (nv * 44 + tt * 36 + hu * 124) + 32.0 / sample_res
    have_valu = bool(ke and "valu_per_segment" in ke)
    have_hbm = bool(ke and ks and "hbm_bytes_per_segment" in ke and "hbm_bytes_per_segment" in ks)
    ginst = ke["valu_per_segment"] * seg_rate / 1e9 if have_valu else None
    hbm_gbs = (ke["hbm_bytes_per_segment"] + ks["hbm_bytes_per_segment"]) * seg / dt / max(n_gpus, 1) / 1e9 if have_hbm else None
    # Top level (round 6): the CU's vector-memory path for divergent fetches — what DESIGN.md section 7.1 finds binding — measured as the busy share of the
    # data-return unit (TD, the busier of the path's two units): TD_TD_BUSY_sum per segment (committed rocprofv3 summary) x the live segment rate of the timed
    # region, over 256 units x 2.4 GHz.  VALU issue (rounds 2-5's top level) stays as the sub-block `valu_issue`.
    have_vmem = bool(ke and "td_busy_cycles_per_segment" in ke)
    unit_gc = UNIT_PEAK_GCYC
    td = ke["td_busy_cycles_per_segment"] * seg_rate / 1e9 if have_vmem else None
    ta = ke["ta_busy_cycles_per_segment"] * seg_rate / 1e9 if (ke and "ta_busy_cycles_per_segment" in ke) else None
    # the counters were collected in the default configuration (two streams per GPU: 256-thread intersect blocks over a 16 KB tile); a one-stream context runs 1024-thread
    # blocks over a 32 KB tile, which make fewer L1 accesses per segment — its launches are not priced with another configuration's cycles per segment
    same_cfg = streams > 1
    ok = have_vmem and not stale and same_cfg
    out = {"bound": "vmem_divergent", "kernel": ext_kernel, "configuration": f"the timed region: {streams} stream(s) per GPU, {n_gpus} GPU(s)",
           "achieved": round(td, 1) if ok else None, "peak": unit_gc, "unit": "G TD-busy cycles/s", "frac": round(td / unit_gc, 4) if ok else None,
           "traffic": round(ke["hbm_bytes_per_segment"] * seg_per_launch) if ke and "hbm_bytes_per_segment" in ke else None,
           "hbm_frac": round(hbm_gbs / HBM_PEAK_GBS, 4) if (have_hbm and not stale) else None,
           "segments_per_launch": round(seg_per_launch), "avg_launch_ms": round(avg_ext, 4), "median_launch_ms": round(r.kernel_time_median("extend"), 4), "launches": n_ext,
           "extend_share_of_step": round(ms_ext / max(world, 1) / (dt * 1e3), 3), "streams_per_gpu": streams,
           "counters_stale": bool(stale), "kernel_source_hash": src_hash, "counters_from": prof_name if prof else None,
           "segments_per_sample": round(S, 3), "per_segment": {"nodes": round(nv, 3), "tritests": round(tt, 3), "hitupd": round(hu, 3)},
           "note": ("bound = the resource the dominant kernel saturates: its CU's vector-memory path for divergent node / triangle fetches (DESIGN.md 7.1: launch time flat from 4 waves "
                    "per SIMD, extra bytes per fetch cost what they weigh).  achieved = TD_TD_BUSY_sum per segment (data-return unit, the busier of the path's two units; committed "
                    "rocprofv3 summary, each --pmc set in a pass of its own) x segments per launch / mean launch duration of the timed region (HIP events on the launch streams); peak = 256 "
                    "units x 2.4 GHz.  `vmem` has both units, the tag lookups per vector-memory instruction and the L1 hit rate; `valu_issue` the vector-instruction issue rate against "
                    "256 CUs x 4 SIMDs x 2.4 GHz / 2; hbm_frac = rocprofv3 HBM bytes of both kernels over the wall time / 8 TB/s (block `hbm`).  SURVEY.md 8(d)'s algorithmic bytes are "
                    "under `algorithmic`: priced in the reference's layout, they never reach HBM for a cache-resident tree and carry no fraction.")}
    if stale:
        out["note"] = ("COUNTERS STALE: " + prof_name + " was measured on other kernel sources (its hash differs from kernel_source_hash): every figure derived from it — frac, "
                       "hbm_frac, the sub-blocks' fractions — is withheld or provisional until scripts/pmc_all.sh has run on this tree.  ") + out["note"]
    if have_vmem and not same_cfg:
        out["note"] = ("ONE stream per GPU: the committed counters are those of the two-stream block configuration, so the top-level fraction is withheld; the directly measured busy "
                       "shares of the kernel alone on the chip are under vmem.alone_on_the_chip.  ") + out["note"]
    if have_vmem:
        out["vmem"] = {"td_busy": {"cycles_per_segment": ke["td_busy_cycles_per_segment"], "achieved": round(td, 1) if same_cfg else None, "frac": round(td / unit_gc, 4) if same_cfg else None},
                       "ta_busy": ({"cycles_per_segment": ke["ta_busy_cycles_per_segment"], "achieved": round(ta, 1) if same_cfg else None, "frac": round(ta / unit_gc, 4) if same_cfg else None} if ta is not None else None),
                       "alone_on_the_chip": {k: ke.get(k) for k in ("ta_busy_frac_alone", "td_busy_frac_alone", "td_waiting_for_cache_frac_alone", "ta_addr_stalled_by_cache_frac_alone")},
                       "tag_lookups_per_vmem_inst": ke.get("tag_lookups_per_vmem_inst"), "l1_hit_rate": ke.get("l1_hit_rate"), "l2_round_trip_cycles": ke.get("l2_round_trip_cycles"),
                       "vmem_rd_per_segment": ke.get("vmem_rd_per_segment"), "provisional": bool(stale)}
    else:
        out["note"] = f"no vector-memory counters in {prof_name}: run scripts/pmc_all.sh on a GPU box.  " + out["note"]
    if have_valu:
        out["valu_issue"] = {"achieved": round(ginst, 1), "peak": VALU_PEAK_GINST, "unit": "Ginst/s", "frac": round(ginst / VALU_PEAK_GINST, 4), "provisional": bool(stale),
                             "valu_insts_per_segment": ke["valu_per_segment"], "salu_insts_per_segment": ke.get("salu_per_segment"), "lane_util": ke.get("lane_util"),
                             "wait_share": ke.get("wait_share"), "issue_stall_share": ke.get("issue_stall_share")}
        if ks and "valu_per_segment" in ks:
            v = (ke["valu_per_segment"] + ks["valu_per_segment"]) * seg / dt / max(n_gpus, 1) / 1e9
            out["chip_valu_issue"] = {"achieved": round(v, 1), "frac": round(v / VALU_PEAK_GINST, 4), "kernel_concurrency": round((ms_ext + ms_sh) / (dt * 1e3) / max(n_gpus, 1), 3),
                                      "note": "both kernels' vector instructions over the wall time of the timed region, per GPU"}
    # ---- measured HBM bytes (north_star: rocprof achieved HBM GB/s against the chip's 8 TB/s), in the contract's form
    hbm = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "note": "rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE (gfx950 correction) per segment from the committed counter summary x the segments of the timed region: both kernels over the "
                   "wall time per GPU; traffic = those bytes per iteration (one intersect + one shading launch); each kernel over its own launches beside it"}
    if have_hbm:
        bps = ke["hbm_bytes_per_segment"] + ks["hbm_bytes_per_segment"]
        hbm.update({"achieved": round(hbm_gbs, 1), "frac": round(hbm_gbs / HBM_PEAK_GBS, 4), "traffic": round(bps * seg_per_launch), "bytes_per_segment": round(bps, 2),
                    "bytes_per_sample": round(bps * S + 32.0 / sample_res, 1),
                    "algorithmic_queue_bytes_per_sample": round(S * 304 + 32.0 / sample_res, 1),
                    ext_kernel: {"bytes_per_segment": ke["hbm_bytes_per_segment"], "achieved": round(ke["hbm_bytes_per_segment"] * seg_rate / 1e9, 1),
                                 "frac": round(ke["hbm_bytes_per_segment"] * seg_rate / 1e9 / HBM_PEAK_GBS, 4)},
                    "k_shade": {"bytes_per_segment": ks["hbm_bytes_per_segment"], "achieved": round(ks["hbm_bytes_per_segment"] * seg_rate_sh / 1e9, 1),
                                "frac": round(ks["hbm_bytes_per_segment"] * seg_rate_sh / 1e9 / HBM_PEAK_GBS, 4)}})
    else:
        hbm.update({"achieved": None, "frac": None, "traffic": None, "note": f"no counter summary in {prof_name}: run scripts/pmc_all.sh on a GPU box"})
    out["hbm"] = hbm
    # ---- the kernels ALONE on the chip (a one-stream pass after the timed region)
    if alone and alone["avg_ext_ms"] > 0:
        rate = alone["seg_per_launch"] / (alone["avg_ext_ms"] * 1e-3)
        al = {"measured_in": alone["from"], "avg_launch_ms": round(alone["avg_ext_ms"], 4), "segments_per_launch": round(alone["seg_per_launch"]), "launches": alone["launches"],
              "algorithmic_GBps": round(b_ext * rate / 1e9, 1)}
        if have_valu:
            al["valu_issue"] = {"achieved": round(ke["valu_per_segment"] * rate / 1e9, 1), "frac": round(ke["valu_per_segment"] * rate / 1e9 / VALU_PEAK_GINST, 4)}
        if have_vmem:      # measured directly: rocprofv3 serialises the kernels of a counter pass, so its busy shares ARE the kernel alone on the chip (in the block configuration of the pass)
            al["td_busy"] = {"frac": ke.get("td_busy_frac_alone"), "ta_busy_frac": ke.get("ta_busy_frac_alone"),
                             "note": "TD_TD_BUSY_sum / (256 x GRBM_GUI_ACTIVE / 8) of the counter pass (two-stream block configuration: 256-thread blocks, 16 KB tile); "
                                     "no per-segment figure is applied to this pass's launches, whose 1024-thread blocks over a 32 KB tile make fewer L1 accesses per segment"}
        if ke and "hbm_bytes_per_segment" in ke:
            al["hbm"] = {"achieved": round(ke["hbm_bytes_per_segment"] * rate / 1e9, 1), "frac": round(ke["hbm_bytes_per_segment"] * rate / 1e9 / HBM_PEAK_GBS, 4)}
        if alone.get("avg_shade_ms", 0) > 0:
            al["k_shade"] = {"avg_launch_ms": round(alone["avg_shade_ms"], 4)}
            if ks and "hbm_bytes_per_segment" in ks:
                gb = ks["hbm_bytes_per_segment"] * alone["seg_per_launch"] / (alone["avg_shade_ms"] * 1e-3) / 1e9
                al["k_shade"]["hbm"] = {"achieved": round(gb, 1), "frac": round(gb / HBM_PEAK_GBS, 4)}
        if alone.get("counters_note"):
            al["counters_note"] = alone["counters_note"]
        out["alone"] = al
    sh = {"kernel": "k_shade", "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "avg_launch_ms": round(avg_sh, 4), "median_launch_ms": round(r.kernel_time_median("shade"), 4),
          "shade_share_of_step": round(ms_sh / max(world, 1) / (dt * 1e3), 3), "algorithmic_bytes_per_segment": 260}
    if ks and "hbm_bytes_per_segment" in ks:
        sh.update({"achieved": round(ks["hbm_bytes_per_segment"] * seg_rate_sh / 1e9, 1), "frac": round(ks["hbm_bytes_per_segment"] * seg_rate_sh / 1e9 / HBM_PEAK_GBS, 4),
                   "bytes_per_segment": ks["hbm_bytes_per_segment"], "traffic": round(ks["hbm_bytes_per_segment"] * seg / max(n_sh, 1)), "lane_util": ks.get("lane_util")})
    out["shade"] = sh
    out["algorithmic"] = {"bytes_per_segment_extend": round(b_ext, 1), "bytes_per_launch_extend": round(b_ext * seg_per_launch), "extend_GBps": round(b_ext * seg_rate / 1e9, 1),
                          "extend_GBps_over_hbm_peak": round(b_ext * seg_rate / 1e9 / HBM_PEAK_GBS, 4),
                          "bytes_per_sample_whole_path": round(b_samp, 1), "whole_path_GBps_per_gpu": round(b_samp * value * 1e6 / 1e9 / max(n_gpus, 1), 1),
                          "whole_path_GBps_over_hbm_peak": round(b_samp * value * 1e6 / 1e9 / max(n_gpus, 1) / HBM_PEAK_GBS, 4),
                          "note": "SURVEY.md 8(d): per segment of rayScene 44 B of queue traffic + 44 B per node visit + 36 B per triangle test + 124 B per hit update; per sample B = S*304 + "
                                  "S*(Nv*44 + Tt*36 + Hu*124) + 32/SAMPLE_RES; S, Nv, Tt, Hu from the statistics pass (equal to the oracle's).  Priced in the reference's buffer layout as if "
                                  "streamed from memory: the device-private BVH is served from LDS / L1 / L2, so the rate is NOT a fraction of a roof and exceeds the HBM peak for a cache-resident tree."}
    return out


def main():
    args = parse()
    if args.spawn and args.gpus > 1 and os.environ.get("WORLD_SIZE") is None:
        import socket
        import subprocess
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
        child = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
                 os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--spawn"]
        raise SystemExit(subprocess.call(child))      # the child's rank 0 prints the JSON line on the inherited stdout
    env_world = os.environ.get("WORLD_SIZE")
    dist_mode = env_world is not None or args.dist          # one process per GPU under torch.distributed.run
    world = int(env_world) if env_world is not None else (1 if args.dist else args.gpus)
    rank = int(os.environ.get("RANK", "0")) if dist_mode else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if dist_mode else 0
    if os.environ.get("PT_BENCH_ONE_GPU") == "1":
        local_rank = 0                              # rehearsal on a one-GPU box: every rank renders on GPU 0
    gloo = dist_mode and args.dist_backend == "gloo"
    if dist_mode and env_world is not None and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: torch.distributed.run --nproc-per-node must equal --gpus")
    K = max(1, args.streams)
    multi = (not dist_mode) and (args.gpus > 1 or args.devices is not None or K > 1) and not args.rehearse_shard

    import numpy as np
    import torch
    import torch.distributed as dist

    import ptimport
    ptimport.load()
    from pathtracer_0_amd import renderer, scenes, shard

    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(local_rank)
    if dist_mode:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if env_world is None:
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        if gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    cdev = torch.device("cpu") if gloo else dev     # where the collectives' tensors live

    cfg = scenes.CONFIGS[args.config]
    W = args.width or cfg["W"]
    H = args.height or cfg["H"]
    sample_res = cfg["sample_res"]
    fps = args.frames_per_step or cfg["spp"] // sample_res
    spp_step = fps * sample_res
    wl = scenes.build(args.config, W, H)
    if args.sky:
        wl.sky = scenes.equirect_sky(args.sky[1], args.sky[0])
        wl.buffers[0] = np.array([0.0, 1.0, -2.6], dtype=np.float32)

    if multi:
        devices = [int(d) for d in args.devices.split(",")] if args.devices else [d for d in range(args.gpus) for _ in range(K)]
        n_gpus = len(set(devices))
        r = renderer.Renderer(W, H, devices=devices)
        shards = len(devices)
    elif dist_mode:
        devices = [local_rank] * K                  # this rank's streams = tile shards rank*K .. rank*K+K-1 of world*K
        r = renderer.Renderer(W, H, devices=devices, first_shard=rank * K, total_shards=world * K)
        n_gpus, shards = world, world * K
    else:
        shard_rank, shard_count = (0, 1) if not args.rehearse_shard else tuple(args.rehearse_shard)
        if args.rehearse_shard and K > 1:           # one GPU's part of a shard_count-GPU run: its K streams = shards rank*K.. of shard_count*K
            r = renderer.Renderer(W, H, devices=[local_rank] * K, first_shard=shard_rank * K, total_shards=shard_count * K)
        else:
            r = renderer.Renderer(W, H, device=local_rank, shard_rank=shard_rank, shard_count=shard_count)
        n_gpus, shards = 1, shard_count
    if args.path_slots:
        r.set_option("path_slots", args.path_slots)
    for name, val in (("lds_budget", args.lds_budget), ("extend_mode", args.extend_mode), ("extend_tpb", args.extend_tpb), ("extend_cache_bytes", args.extend_cache),
                      ("refill_min", args.refill_min), ("none_min", args.none_min), ("extend_blocks_per_cu", args.extend_blocks_per_cu), ("inner_keep_eighths", args.inner_keep), ("bfs_nodes", args.bfs_nodes), ("stack_mode", args.stack_mode),
                      ("asm_loop", args.asm_loop), ("asm_tpb", args.asm_tpb), ("asm_node_layout", args.asm_node_layout), ("asm_root_cull", args.asm_root_cull), ("cu_partition", args.cu_partition)):
        if val is not None:
            r.set_option(name, val)
    if args.contract == "fast":
        r.set_option("numeric_contract", 1)
    r.load_workload(wl)
    r.reset_frame()

    unshard = None
    if multi:
        full_view = lambda ptr: torch.as_tensor(shard._DevArray(ptr, (H, W, 4)), device=torch.device("cuda", devices[0]))      # noqa: E731
        collect = lambda age: full_view(r.gather_image(age))                                                                     # noqa: E731
        collect(0)                              # untimed: RCCL builds its communicators and rings on the first collective
        r.synchronize()
    elif dist_mode:
        # this rank's block: its K packed shard accumulators (world 1: the group holds the whole image and hands it over un-tiled)
        n_block = K * renderer.shard_slots(W, H, world * K) if world > 1 else W * H
        unshard = shard.Unsharder(W, H, world * K if world > 1 else 1, renderer.shard_map, dev, renderer=r, ranks=world, sync_before_unshard=True)

        def collect(age):                       # the library completes the image and packs this rank's block, RCCL gathers the blocks
            # the block of the previous image must have left in its collective before the library packs the next one into the same buffer
            # (dist.gather returns with torch's stream waiting for the RCCL kernel, not the host; ranks ahead of the root could overtake it)
            torch.cuda.current_stream(dev).synchronize()
            packed = torch.as_tensor(shard._DevArray(r.gather_image(age), (n_block, 4)), device=dev)
            r.stream_wait()                     # the block was written on the library's stream; torch's collective runs on another
            if gloo:                            # rehearsal: the same gather on CPU tensors, the gathered blocks go back to the GPU for pt_unshard
                packed = packed.cpu()
                unshard.to_device = dev
            return shard.gather_frame(packed, unshard, dst=0, force_collective=True)
        collect(0)                              # untimed: RCCL builds its communicator and rings on the first collective, whatever --warmup is
        torch.cuda.synchronize(dev)
    else:
        if args.rehearse_shard and K > 1:
            collect = lambda age: r.gather_image(age)                                             # noqa: E731  (the packed block, as a pointer)
        elif args.rehearse_shard:
            collect = lambda age: (r.finish_image(age), shard.frame_tensor(r, dev, age))[1]      # noqa: E731
        else:
            collect = lambda age: torch.as_tensor(shard._DevArray(r.gather_image(age), (H, W, 4)), device=dev)      # noqa: E731

    MAX_BATCH = args.max_batch
    # One step = one image: reset, spp_step samples per pixel, ONE framebuffer gather.  Consecutive steps overlap on the GPU:
    # a step's last paths finish underneath the next step's first ones (pt_render_batch_async / pt_next_image), and its
    # image is gathered while the next one renders.  Every step's work and gather lie inside the timed region; the fence
    # completes everything that is still in flight.
    pipeline = shard.StepPipeline(r, unshard, dev, lag=2, collect=collect)      # an image is gathered two steps after it was submitted

    def submit():
        done = 0
        while done < fps:
            n = min(MAX_BATCH, fps - done)
            first = 1 + done
            seeds = [scenes.frame_seed(f) for f in range(first, first + n)]
            if args.sync:
                r.render_batch(first, seeds)
            else:
                r.render_batch_async(first, seeds)
            done += n

    def step():
        if args.sync:
            r.reset_frame()
            submit()
            return collect(0)
        return pipeline.step(submit)

    def drain():
        imgs = pipeline.drain()
        return imgs[-1] if imgs else None

    def fence():
        r.synchronize()
        if dist_mode:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- untimed: statistics pass for the roofline (counters are deterministic per scene + seeds) ----
    stats = None
    if not args.no_roofline:
        r.set_option("count_stats", 1)
        r.reset_counters()
        nstat = min(fps, 2)
        r.render_batch(1, [scenes.frame_seed(f) for f in range(1, 1 + nstat)])
        r.synchronize()
        stats = r.counters()
        if dist_mode and world > 1:             # whole-image totals
            t = torch.tensor([float(stats[k]) for k in renderer.COUNTERS], dtype=torch.float64, device=cdev)
            dist.all_reduce(t)
            stats = dict(zip(renderer.COUNTERS, [int(x) for x in t.tolist()]))
        r.set_option("count_stats", 0)
        r.reset_frame()

    # untimed set-up: one pass sizes and allocates the path pool and the per-frame rings (several GB of hipMalloc), whatever --warmup is
    step()
    drain()
    fence()
    for _ in range(args.warmup):
        step()
    drain()
    fence()
    r.reset_counters()
    if not args.no_roofline:
        r.set_timing(True)
    t0 = time.perf_counter()
    full = None
    for _ in range(args.steps):
        full = step()
    full = drain() if not args.sync else full
    fence()
    dt = time.perf_counter() - t0
    r.set_timing(False)
    if dist_mode:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    samples = float(W) * H * spp_step * args.steps
    if args.rehearse_shard:
        own = [renderer.shard_map(W, H, shard_rank * K + k, shard_count * K) for k in range(K)] if K > 1 else [renderer.shard_map(W, H, shard_rank, shard_count)]
        samples = float(sum((m >= 0).sum() for m in own)) * spp_step * args.steps
    value = samples / dt / 1e6

    per_gpu = f"{shards // max(n_gpus, 1)} independent wavefront stream(s) per GPU"
    if multi:
        how = (f"ONE process, pt_create_multi on devices {devices} ({per_gpu}): " +
               ("device copies between the streams of a GPU, RCCL ncclGather across GPUs, un-tiling on device " if n_gpus > 1 else "device copies between the streams, un-tiling on device ") + str(devices[0]))
    elif dist_mode:
        how = (f"one process per GPU, {per_gpu} (pt_create_multi_part), torch.distributed {'nccl (= RCCL)' if not gloo else 'gloo (REHEARSAL on CPU tensors)'} "
               f"dist.gather of the packed blocks on rank 0, pt_unshard there; world {world}")
    else:
        how = "one GPU, one stream, no collective"
    out = {
        "metric": METRIC, "value": round(value, 3), "unit": "Msamples/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic", "contract": args.contract, "hip_runtime": renderer.hip_runtime_info(),
        "config": {"workload": f"{args.config}: {W}x{H}, {cfg['bounces']}-bounce, {spp_step} spp/step ({fps} frames x SAMPLE_RES {sample_res}), "
                               f"{wl.info['triangles']} triangles / {wl.info['objects']} BVHs" + (f", {args.sky[0]}x{args.sky[1]} equirect sky" if args.sky else "") + f", tile-sharded over {shards} shard(s) on {n_gpus} GPU(s), 1 framebuffer gather per step",
                   "width": W, "height": H, "max_bounces": cfg["bounces"], "spp_per_step": spp_step, "triangles": wl.info["triangles"], "multi_gpu": how},
    }
    if gloo or os.environ.get("PT_BENCH_ONE_GPU") == "1":
        out["rehearsal"] = "process-per-GPU form rehearsed with a gloo collective and/or all ranks on one GPU: not a scaling measurement"
    if args.rehearse_shard:
        out["rehearsal"] = f"GPU {shard_rank} of {shard_count} alone ({K} stream(s)): value is THIS GPU's share of the image at its own rate, not a multi-GPU measurement"
    if stats is not None:
        if dist_mode and world > 1:             # launches and device time of all ranks, like the multi-GPU context reports them
            acc = []
            for k in ("extend", "shade"):
                n_k, ms_k = r.kernel_time(k)
                acc += [float(n_k), ms_k]
            t = torch.tensor(acc, dtype=torch.float64, device=cdev)
            dist.all_reduce(t)
            tot = t.tolist()
            med = {k: r.kernel_time_median(k) for k in ("extend", "shade")}

            class _All:                         # the same three calls roofline_block makes on a renderer
                def kernel_time(self, k):
                    i = 0 if k == "extend" else 2
                    return int(tot[i]), tot[i + 1]

                def kernel_time_median(self, k):
                    return med[k]
            src = _All()
        else:
            src = r
        try:                                    # which intersect kernel ran: the hand-written one (pt_extend_gfx950.s) or the compiled k_extend_persist
            r.set_option("query_asm_launches_above", 0)
            ext_kernel = "pt_extend_asm"
        except renderer.PtError:
            ext_kernel = "k_extend_persist"
        alone = None
        if multi and n_gpus == 1 and shards > 1 and rank == 0 and not args.no_alone_pass:
            # the kernels ALONE on the chip: two steps of the same workload on a one-stream context, after the timed region
            r1 = renderer.Renderer(W, H, device=devices[0])
            for name, val in (("extend_mode", args.extend_mode), ("extend_cache_bytes", args.extend_cache), ("refill_min", args.refill_min), ("none_min", args.none_min),
                              ("asm_loop", args.asm_loop), ("asm_tpb", args.asm_tpb), ("asm_node_layout", args.asm_node_layout), ("asm_root_cull", args.asm_root_cull), ("cu_partition", args.cu_partition), ("path_slots", args.path_slots), ("numeric_contract", 1 if args.contract == "fast" else None)):
                if val is not None:
                    r1.set_option(name, val)
            r1.load_workload(wl); r1.reset_frame()
            seeds_a = [scenes.frame_seed(f) for f in range(1, 1 + min(fps, MAX_BATCH))]
            r1.render_batch_async(1, seeds_a); r1.synchronize(); r1.reset_counters()      # untimed: pool allocation
            r1.set_timing(True)
            for _ in range(2):
                r1.render_batch_async(1, seeds_a)
            r1.synchronize(); r1.set_timing(False)
            na, msa = r1.kernel_time("extend"); ns, mss = r1.kernel_time("shade")
            S_ = stats["segments"] / max(stats["samples"], 1)
            alone = {"avg_ext_ms": msa / max(na, 1), "avg_shade_ms": mss / max(ns, 1), "seg_per_launch": S_ * 2.0 * len(seeds_a) * sample_res * W * H / max(na, 1), "launches": na,
                     "from": f"a one-stream pass after the timed region: 2 x {len(seeds_a)} frames of the same workload, the kernels alone on the chip",
                     "counters_note": "per-segment counter figures are those of the committed summary (two streams per GPU, 256-thread intersect blocks); alone on its GPU the intersect "
                                      "kernel runs 1024-thread blocks, whose counters were last compared in round 3 (about 1 % in VALU, 4 % in HBM bytes per segment: profiles/r03_m_pmc_C3_one_stream.txt, an older kernel generation); `vmem.alone_on_the_chip` of the top level holds the busy shares rocprofv3 measured directly with each kernel alone on the chip (a counter pass serialises the kernels)"}
            r1.close()
        out["roofline"] = roofline_block(args, src, stats, samples, shards if not args.rehearse_shard else 1, dt, value, sample_res, n_gpus=n_gpus, ext_kernel=ext_kernel, alone=alone)
        out["hbm_frac"] = out["roofline"]["hbm_frac"]      # north_star's figure at the top of the line: measured HBM bytes of both kernels / wall time / 8 TB/s

    if rank == 0 and full is not None and hasattr(full, "cpu") and not args.rehearse_shard:
        # SURVEY.md 8(d) ends the clock at the completion of pt_read_frame; `value` ends at the gathered image in HBM (results resident, like the inputs).
        # The read-back of one image over PCIe, measured here after the timed region (pinned host memory), and the rate with one read-back per step added:
        host = torch.empty(full.shape, dtype=full.dtype, pin_memory=True)
        torch.cuda.synchronize(full.device)
        tr = time.perf_counter()
        for _ in range(3):
            host.copy_(full, non_blocking=True)
        torch.cuda.synchronize(full.device)
        rb = (time.perf_counter() - tr) / 3.0
        out["readback"] = {"ms_per_image": round(rb * 1e3, 3), "bytes": int(full.numel() * full.element_size()),
                           "value_with_one_readback_per_step": round(samples / (dt + rb * args.steps) / 1e6, 3),
                           "note": "value ends at the gathered image in device memory; pt_read_frame adds this copy (not overlapped here: an upper bound on its cost)"}
    if rank == 0 and n_gpus == 1 and not dist_mode and not args.rehearse_shard and not args.no_cpu_baseline:
        # the reference has no CPU render path: the timed CPU baseline is the oracle ("port")
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle
        cores, core_info = host_cores()         # every core this process may use (affinity mask, cgroup quota)
        sc = oracle.Scene.from_workload(wl)
        xs = ys = 2 if W * H > 3000000 else 1
        buf = np.zeros((H, W, 4), dtype=np.float32)
        # bounded sample: frame 1, then as many further frames of the same workload as fit into ~12 s of host time (at most 16)
        tc = time.perf_counter()
        _, ocnt = oracle.render(sc, W, H, 1, scenes.frame_seed(1), buf, nthreads=cores, xs=xs, ys=ys)
        t1 = time.perf_counter() - tc
        csamp, nfr = float(ocnt[4]), 1
        more = max(0, min(15, int(12.0 / max(t1, 1e-3)) - 1))
        for f in range(2, 2 + more):
            _, ocnt = oracle.render(sc, W, H, f, scenes.frame_seed(f), buf, nthreads=cores, xs=xs, ys=ys)
            csamp += float(ocnt[4]); nfr += 1
        tcpu = time.perf_counter() - tc
        # the metric's second half: per-pixel RMSE of the same frames, HIP path vs oracle (the oracle is the checker here, never the product)
        r.reset_frame()
        r.render_batch(1, [scenes.frame_seed(f) for f in range(1, nfr + 1)])
        gpu = r.read_frame()
        ia, ib = gpu[::ys, ::xs, :3] / float(nfr), buf[::ys, ::xs, :3] / float(nfr)
        diff = (ia.astype(np.float64) - ib.astype(np.float64))
        same = bool(np.array_equal(gpu[::ys, ::xs], buf[::ys, ::xs], equal_nan=True))
        out["parity"] = {"rmse_vs_oracle": float(np.sqrt(np.nanmean(diff ** 2))), "bit_identical": same, "tolerance": 1e-3, "contract": args.contract,
                         "sample": f"frames 1..{nfr} of the timed workload ({nfr * sample_res} spp), every {xs}th pixel in x and y"
                                   + (" — the relaxed contract is not bit-identical by design; at the workload's full spp its RMSE is what tests/test_gpu_parity.py::test_fast_contract_rmse asserts"
                                      if args.contract == "fast" else "")}
        out["cpu_baseline"] = {"value": round(csamp / tcpu / 1e6, 3), "unit": "Msamples/s", "cores": cores, **core_info, "kind": "port",
                               "sample": f"oracle (C++ restatement of frag.glsl), frames 1..{nfr} ({sample_res} spp each) of the same workload at every {xs}th pixel in x and y: "
                                         f"{int(csamp)} samples in {tcpu:.2f} s"}
    elif rank == 0 and (multi or dist_mode) and full is not None and not args.no_cpu_baseline:
        # N > 1: the gathered image of the last step against a one-GPU render of the same step (K10: bit-identical for every shard count)
        r1 = renderer.Renderer(W, H, device=(devices[0] if multi else local_rank))
        r1.load_workload(wl); r1.reset_frame()
        done = 0
        while done < fps:
            n = min(MAX_BATCH, fps - done)
            r1.render_batch(1 + done, [scenes.frame_seed(f) for f in range(1 + done, 1 + done + n)])
            done += n
        ref = r1.read_frame(); r1.close()
        got = full.cpu().numpy()
        out["parity"] = {"gathered_image_bit_identical_to_one_gpu_render": bool(np.array_equal(got, ref, equal_nan=True)),
                         "sample": f"last step's gathered {W}x{H} image ({spp_step} spp) vs the same step rendered unsharded on one GPU"}
    if rank == 0:
        print(json.dumps(out), flush=True)      # before any teardown: the line is the product of the run
    r.close()
    if dist_mode:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
