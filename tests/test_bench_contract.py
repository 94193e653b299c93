"""CPU: the bench line committed under profiles/ carries every key of the driver's contract (and bench.py parses its flags without a GPU)."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_keys():
    latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default_c3.json")))[-1]
    d = json.load(open(latest))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == base["metric"] and d["unit"] == "Msamples/s" and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    if os.path.basename(latest) >= "r06":
        # round 6: the top level is the resource DESIGN.md 7.1 finds binding — the CU's vector-memory path for divergent fetches — as the busy share of its data-return
        # unit in the timed configuration (TD_TD_BUSY_sum per segment from the committed counter summary of THIS tree's kernels x the live segment rate / 256 units x 2.4 GHz);
        # VALU issue is the sub-block `valu_issue`; `hbm_frac`, `hbm`, `alone`, `shade`, `algorithmic` as in round 5
        assert r["bound"] == "vmem_divergent" and 0 < r["frac"] <= 1 and r["counters_stale"] is False and r["peak"] == 614.4
        prof = json.load(open(os.path.join(ROOT, r["counters_from"])))
        pk = prof["kernels"][r["kernel"]]
        vm = r["vmem"]
        assert abs(pk["td_busy_cycles_per_segment"] - vm["td_busy"]["cycles_per_segment"]) < 1e-9 and vm["provisional"] is False
        assert abs(r["achieved"] - vm["td_busy"]["cycles_per_segment"] * r["segments_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 2e-2
        assert r["frac"] == vm["td_busy"]["frac"] and 0 < vm["ta_busy"]["frac"] <= 1 and 0 < vm["alone_on_the_chip"]["td_busy_frac_alone"] <= 1 and 0 < vm["alone_on_the_chip"]["ta_busy_frac_alone"] <= 1
        assert vm["tag_lookups_per_vmem_inst"] > 1 and 0 < vm["l1_hit_rate"] < 1
        v = r["valu_issue"]
        assert abs(pk["valu_per_segment"] - v["valu_insts_per_segment"]) < 1e-9 and 0 < v["lane_util"] <= 1 and 0 < v["frac"] <= 1 and v["unit"] == "Ginst/s"
        assert abs(v["achieved"] - v["valu_insts_per_segment"] * r["segments_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) / v["achieved"] < 2e-2
        assert abs(r["traffic"] - pk["hbm_bytes_per_segment"] * r["segments_per_launch"]) / r["traffic"] < 1e-2
        h = r["hbm"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in h, k
        assert h["bound"] == "hbm" and h["unit"] == "GB/s" and h["peak"] == 8000.0 and 0 < h["frac"] <= 1 and abs(h["frac"] - h["achieved"] / 8000.0) < 1e-3
        assert r["hbm_frac"] == h["frac"] == d["hbm_frac"]
        assert 0 < h[r["kernel"]]["frac"] <= 1 and 0 < h["k_shade"]["frac"] <= 1 and 0 < r["chip_valu_issue"]["frac"] <= 1
        assert "measured_in" in r["alone"] and 0 < r["alone"]["valu_issue"]["frac"] <= 1 and 0 < r["alone"]["hbm"]["frac"] <= 1
        assert r["alone"]["td_busy"]["frac"] == pk["td_busy_frac_alone"] and 0 < pk["td_busy_frac_alone"] <= 1
        assert r["shade"]["bound"] == "hbm" and 0 < r["shade"]["frac"] <= 1

        def no_frac_key(o):
            return all(k != "frac" and no_frac_key(v) for k, v in o.items()) if isinstance(o, dict) else True
        assert no_frac_key(r["algorithmic"]) and r["algorithmic"]["extend_GBps"] > 0
        assert d["hip_runtime"]["version"] and d["hip_runtime"]["runtimes_mapped"] == 1
        assert d["cpu_baseline"]["cores"] <= d["cpu_baseline"]["cores_available"] and d["readback"]["ms_per_image"] > 0
    elif os.path.basename(latest) >= "r05":
        # round 5: every fraction of the line is a fraction of a roof the hardware can reach.  Top level = the dominant kernel against the resource that binds it
        # (vector-instruction issue), `hbm_frac` (in the block and at the top of the line) = measured HBM bytes of both kernels over the wall time / 8 TB/s,
        # `hbm` = the same in the contract's bound / achieved / peak / unit / frac / traffic form; SURVEY.md 8(d)'s algorithmic bytes sit under `algorithmic` without a fraction
        assert r["bound"] == "valu_issue" and r["unit"] == "Ginst/s" and 0 < r["frac"] <= 1 and r["counters_stale"] is False
        prof = json.load(open(os.path.join(ROOT, r["counters_from"])))
        pk = prof["kernels"][r["kernel"]]
        assert abs(pk["valu_per_segment"] - r["valu_insts_per_segment"]) < 1e-9 and 0 < r["lane_util"] <= 1
        assert abs(r["achieved"] - r["valu_insts_per_segment"] * r["segments_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 2e-2
        assert abs(r["traffic"] - pk["hbm_bytes_per_segment"] * r["segments_per_launch"]) / r["traffic"] < 1e-2
        h = r["hbm"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in h, k
        assert h["bound"] == "hbm" and h["unit"] == "GB/s" and h["peak"] == 8000.0 and 0 < h["frac"] <= 1 and abs(h["frac"] - h["achieved"] / 8000.0) < 1e-3
        assert r["hbm_frac"] == h["frac"] == d["hbm_frac"]
        assert 0 < h[r["kernel"]]["frac"] <= 1 and 0 < h["k_shade"]["frac"] <= 1 and 0 < r["chip_valu_issue"]["frac"] <= 1
        assert "measured_in" in r["alone"] and 0 < r["alone"]["valu_issue"]["frac"] <= 1 and 0 < r["alone"]["hbm"]["frac"] <= 1
        assert r["shade"]["bound"] == "hbm" and 0 < r["shade"]["frac"] <= 1

        def no_frac_key(o):                       # the algorithmic figure is bookkeeping: nothing under it may read as a roofline fraction
            return all(k != "frac" and no_frac_key(v) for k, v in o.items()) if isinstance(o, dict) else True
        assert no_frac_key(r["algorithmic"]) and r["algorithmic"]["extend_GBps"] > 0
        assert d["hip_runtime"]["version"] and d["hip_runtime"]["runtimes_mapped"] == 1
        assert d["cpu_baseline"]["cores"] <= d["cpu_baseline"]["cores_available"] and d["readback"]["ms_per_image"] > 0
    elif os.path.basename(latest) >= "r04":
        # round 4: the top level is the TIMED configuration against the HBM roof in SURVEY.md 8(d)'s terms (algorithmic bytes per launch / live launch
        # duration; priced in the reference's layout, so a cache-resident tree may exceed 1), `traffic` and `hbm` are the measured bytes (rocprofv3
        # FETCH x 2 + WRITE of both kernels over the wall time: north_star's figure), VALU issue and the kernels alone on the chip sit under their own keys
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["frac"] > 0 and r["counters_stale"] is False
        assert abs(r["achieved"] - r["algorithmic_bytes_per_segment"] * r["segments_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 2e-2
        prof = json.load(open(os.path.join(ROOT, r["counters_from"])))
        pk = prof["kernels"][r["kernel"]]
        assert abs(r["traffic"] - pk["hbm_bytes_per_segment"] * r["segments_per_launch"]) / r["traffic"] < 1e-2
        assert 0 < r["hbm"]["frac"] <= 1 and abs(r["hbm"]["frac"] - r["hbm"]["achieved"] / 8000.0) < 1e-3
        assert 0 < r["hbm"][r["kernel"]]["frac"] <= 1 and 0 < r["hbm"]["k_shade"]["frac"] <= 1
        assert 0 < r["valu_issue"]["frac"] <= 1 and abs(pk["valu_per_segment"] - r["valu_issue"]["valu_insts_per_segment"]) < 1e-9 and 0 < r["valu_issue"]["lane_util"] <= 1
        assert "measured_in" in r["alone"] and 0 < r["alone"]["valu_issue"]["frac"] <= 1 and 0 < r["alone"]["hbm"]["frac"] <= 1
        assert r["shade"]["bound"] == "hbm" and 0 < r["shade"]["frac"] <= 1
        assert d["hip_runtime"]["version"] and d["hip_runtime"]["runtimes_mapped"] == 1
        assert d["cpu_baseline"]["cores"] <= d["cpu_baseline"]["cores_available"] and d["readback"]["ms_per_image"] > 0
    elif os.path.basename(latest) >= "r02":
        # every fraction is a fraction of a roof the kernel can actually hit: the intersect kernel against VALU issue, its measured HBM bytes and
        # the shading kernel's against the 8 TB/s peak; the counters behind them are a committed rocprofv3 summary
        assert r["bound"] == "valu_issue" and 0 < r["frac"] <= 1 and 0 < r["hbm"]["frac"] <= 1 and 0 < r["shade"]["frac"] <= 1 and r["shade"]["bound"] == "hbm"
        prof = json.load(open(os.path.join(ROOT, r["counters_from"])))
        assert abs(prof["kernels"][r.get("kernel", "k_extend_persist")]["valu_per_segment"] - r["valu_insts_per_segment"]) < 1e-9
        assert abs(r["achieved"] - r["valu_insts_per_segment"] * r["segments_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 2e-2
        if os.path.basename(latest) >= "r03":
            # the top-level figure is the kernel ALONE on the chip; what one stream's launch reaches beside the other streams' kernels sits under in_run
            assert "measured_in" in r and 0 < r["in_run"]["frac"] <= 1 and r["in_run"]["streams_per_gpu"] >= 1 and r["counters_stale"] is False
            assert d["hip_runtime"]["version"] and d["hip_runtime"]["runtimes_mapped"] == 1
    else:
        assert r["bound"] in ("hbm", "mfma")
    assert d["parity"]["bit_identical"] is True and d["parity"]["rmse_vs_oracle"] <= d["parity"]["tolerance"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["value"] > 0
    assert abs(d["value"] - d["config"]["width"] * d["config"]["height"] * d["config"]["spp_per_step"] / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3


def test_bench_parses_its_flags_without_a_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "--gpus" in out.stdout and "--steps" in out.stdout and "--warmup" in out.stdout


def test_host_cores_respects_the_cgroup_quota(tmp_path, monkeypatch):
    """bench.py's cpu_baseline uses every core the process may use: the smallest of os.cpu_count(), the affinity mask and the cgroup CPU quota"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    n, info = b.host_cores()
    assert 1 <= n <= info["cores_available"] and n <= info.get("affinity", n)
    if "cgroup_cpu_quota" in info:
        assert n <= max(1, int(info["cgroup_cpu_quota"] + 0.5))
