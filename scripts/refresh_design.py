#!/usr/bin/env python3
"""Rewrites the measured figures of DESIGN.md §5 / §6 from the committed round-5 profiles (after scripts/collect_profiles.sh <tag> r05):
    refresh_design.py <gpurun tag of the pass> "<earlier driver-command values, comma separated>" """
import json
import re
import sys

tag, earlier = sys.argv[1], sys.argv[2]
p = 'DESIGN.md'; s = open(p).read()
d = json.load(open('profiles/r05_bench_default_c3.json')); r = d['roofline']
cfg = {c: json.load(open(f'profiles/r05_bench_{c}.json'))['value'] for c in ('C2', 'C4', 'C5', 'C6')}
one = json.load(open('profiles/r05_bench_c3_one_stream.json'))['value']; tr = json.load(open('profiles/r05_bench_torchrun_n1.json'))['value']
nfr = re.search(r"frames 1\.\.(\d+)", d['parity']['sample']).group(1)
fb = {}
for l in open('profiles/r05_fallback_paths.txt'):
    m = re.match(r"(.*?)\s+([\d.]+) Msamples/s", l)
    if m:
        fb[m.group(1).strip()] = float(m.group(2))
hf = next(v for k, v in fb.items() if k.startswith('height field')); hfc = fb['... forced onto the compiled kernel (extend_mode 1)']
d3 = fb['C3, RAYTRACING = 0 (directDiffuse)']; d5 = fb['C5, RAYTRACING = 0 (directDiffuse + thickness probes)']
c4a = fb['C4, hand-written kernel']; c4c = fb['C4, compiled kernel (extend_mode 1)']
reh = [float(re.search(r"ms/step ([\d.]+)", l).group(1)) for l in open('profiles/r05_shard_rehearsals.txt') if l.startswith('two streams')]
n_pass = len(earlier.split(',')) + 1
a = s.index('**Results.** Driver-timed: round 1 1511'); b = s.index('**Rooflines** (`roofline` block')
s = s[:a] + f'''**Results.** Driver-timed: round 1 1511, round 2 1792.1, round 3 2045.1, round 4 **2172.6** Msamples/s (`BENCH_r0*.json`). Round 5, the driver's command
(`python3 bench.py --gpus 1 --steps 20 --warmup 5`) in {n_pass} measurement passes on boxes of the pool as the kernels changed: {earlier.replace(',', ', ')} (the first of them:
`profiles/r05_bench_default_c3_earlier_box.json`) and, on the final kernels, **{d['value']:.1f}** Msamples/s (gpurun {tag}: `profiles/r05_bench_default_c3.json`, {d['ms_per_step']:.1f} ms per 256-spp step; one stream
{one:.0f}; under `torch.distributed.run` {tr:.0f}; a 150-step soak 2189.6, `r05_l_soak.txt`); frames 1..{nfr} bit-identical to the oracle at full resolution, the oracle on the 16 host cores the box
grants (of 256) {d['cpu_baseline']['value']:.1f} Msamples/s. Boxes of the pool differ by up to 8 % (the passes span 2.7 % on kernels that A/B within ±0.5 % of each other on C3), so every comparison in
`profiles/` is made within ONE gpurun call. C3 itself moved little this round (the round went into C6, large trees, the direct-diffuse mode, textured materials, the fetch-path measurement
of §7 and the review's correctness items); same final build, `bench.py --config` (3 steps, each line with its `parity` block — bit-identical — and `cpu_baseline`): C2 {cfg['C2']:.0f}, C4 {cfg['C4']:.0f},
C5 {cfg['C5']:.0f}, **C6 {cfg['C6']:.0f}** (round 4: 620) Msamples/s. `profiles/r05_fallback_paths.txt` — what moved onto the hand-written kernel: a 1 M-triangle height field (2 M nodes) **{hf:.0f}
Msamples/s against {hfc:.0f} on the compiled kernel it ran on until this round**; directDiffuse **{d3 / 1000:.1f} G (C3; was 10.4) / {d5 / 1000:.1f} G (C5 with thickness probes; was 8.0)**; C4 forced onto the
compiled kernel: {c4c:.0f} against {c4a:.0f} ({(c4c / c4a - 1) * 100:.0f} %). 9600 random scenes and 4200 random API sequences over every switch incl. the new ones: 0 mismatches (`r05_i_fuzz.txt`).

''' + s[b:]
a = s.index('* **Top level = the dominant kernel (`pt_extend_asm`) in the TIMED configuration against vector-instruction issue**'); b = s.index('* **`algorithmic`** = SURVEY.md')
s = s[:a] + f'''* **Top level = the dominant kernel (`pt_extend_asm`) in the TIMED configuration against vector-instruction issue** (`bound: "valu_issue"`): {r['valu_insts_per_segment']:.1f} vector instructions
  per segment (lane_util {r['lane_util']:.3f}) × {r['segments_per_launch'] / 1e6:.2f} M segments per launch ÷ {r['avg_launch_ms']:.3f} ms = {r['achieved']:.0f} G wave-instructions/s = **{r['frac']:.3f}** of 256 CUs × 4 SIMDs × 2.4 GHz ÷ 2 cycles;
  alone on the chip {r['alone']['valu_issue']['frac']:.2f} ({r['alone']['avg_launch_ms']:.3f} ms per launch), both kernels over the wall time {r['chip_valu_issue']['frac']:.2f}. `traffic` = the kernel's measured HBM bytes per launch ({r['hbm'][r['kernel']]['bytes_per_segment']:.1f} B per
  segment: 32 B of ray read + 16 B of hit record written + node misses). §7.1: what binds the kernel is its CU's vector-memory path for divergent fetches, for which rocprofv3
  offers no single counter with a peak; VALU issue is the measured roof it sits closest to.
* **`hbm_frac`** (in the block and at the top of the line) = what north_star asks for: rocprofv3 bytes (FETCH × 2 + WRITE, the guide's gfx950 correction) of BOTH kernels over the
  wall time of the timed region, per GPU: {r['hbm']['bytes_per_segment']:.0f} B per segment × 3.925 segments per sample × {d['value']:.0f} M samples/s = **{r['hbm']['achieved'] / 1000:.2f} TB/s = {r['hbm']['frac']:.3f} of the peak**; `hbm` carries it in the
  contract's bound / achieved / peak / unit / frac / traffic form, and per kernel over its own launches: `pt_extend_asm` {r['hbm'][r['kernel']]['frac']:.3f}, `k_shade` {r['hbm']['k_shade']['frac']:.2f} ({r['hbm']['k_shade']['bytes_per_segment']:.0f} B per segment; alone on
  the chip {r['alone']['k_shade']['hbm']['frac']:.2f}). Queue-only algorithmic bytes are 304 B per segment: the measured {r['hbm']['bytes_per_segment']:.0f} B are BELOW that because the index stack travels as 3-bit codes, incLight only when it is
  not zero, and the sample sums only when a sample ends (§2).
''' + s[b:]
a = s.index('* **`algorithmic`** = SURVEY.md'); b = s.index('* The rocprofv3 kernel trace of the same command')
s = s[:a] + f'''* **`algorithmic`** = SURVEY.md §8(d)'s bytes, kept as bookkeeping with no fraction: per `rayScene` segment 44 B of queue traffic + 44 B per node visit + 36 B per triangle test + 124 B
  per hit update in the reference's buffer layout — on C3 {r['algorithmic']['bytes_per_segment_extend']:.0f} B per segment, {r['algorithmic']['extend_GBps'] / 1000:.1f} TB/s over the launch durations ({r['algorithmic']['extend_GBps_over_hbm_peak']:.2f} × the HBM peak: the 205 KB tree is served from LDS, L1 and L2).
''' + s[b:]
s = re.sub(r"serially: \d+ instead of \d+ Msamples/s\.", f"serially: {d['readback']['value_with_one_readback_per_step']:.0f} instead of {d['value']:.0f} Msamples/s.", s)
rows = [l.rstrip().split(',') for l in open('profiles/r05_kernel_trace_stats_c3.txt') if l.startswith(('pt_extend_asm', 'k_shade<3, false'))]
te, ts = float(rows[0][-4]) / 1000, float(rows[1][-4]) / 1000
m = re.search(r'bench line of the traced run: .*"avg_launch_ms": ([\d.]+).*shade avg ([\d.]+)', open('profiles/r05_kernel_trace_stats_c3.txt').read())
s = re.sub(r"gives \d\.\d+ ms per `pt_extend_asm` launch over all 1790 launches of the process, the HIP events of the same run \d\.\d+ ms over the 1090 launches",
           f"gives {te:.3f} ms per `pt_extend_asm` launch over all 1790 launches of the process, the HIP events of the same run {float(m.group(1)):.3f} ms over the 1090 launches", s)
s = re.sub(r"with their shorter tails\); `k_shade` \d\.\d+ ms against \d\.\d+\.", f"with their shorter tails); `k_shade` {ts:.3f} ms against {float(m.group(2)):.3f}.", s)
i = s.index("collective) are in `profiles/r05_shard_rehearsals.txt`:"); j = s.index("before the gather.", i) + len("before the gather.")
s = s[:i] + (f"collective) are in `profiles/r05_shard_rehearsals.txt`: {reh[0]:.1f} / {reh[1]:.1f} / {reh[2]:.1f} / {reh[3]:.1f} ms per step for 1 / 2 / 4 / 8 GPUs (the round's other passes: 240.1-247.2 / 120.6-124.3 / "
             f"62.2-64.3 / 32.9-34.0; round 4: 240.5 / 122.5 / 63.5 / 33.5; round 3: 260.0 / 129.5 / 66.8 / 35.0): {reh[0] / reh[3]:.1f} × before the gather.") + s[j:]
open(p, 'w').write(s)
print("DESIGN.md refreshed from", tag, d['value'], cfg, hf, d3, d5)
