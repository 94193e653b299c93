#!/usr/bin/env python3
"""Host CPU time of a bench.py run (the multi-stream context's worker threads poll their streams): host_cpu.py [bench flags]
prints wall seconds, user + system CPU seconds of the child process and the cores that makes."""
import os
import resource
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
t0 = time.time()
p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + sys.argv[1:], capture_output=True, text=True)
wall = time.time() - t0
ru = resource.getrusage(resource.RUSAGE_CHILDREN)
line = [l for l in p.stdout.splitlines() if l.startswith("{")]
print(f"bench.py {' '.join(sys.argv[1:])}: wall {wall:.1f} s, user {ru.ru_utime:.1f} s, system {ru.ru_stime:.1f} s -> {(ru.ru_utime + ru.ru_stime) / wall:.2f} cores on average (whole process: imports, scene build, set-up, timed region)")
if line:
    import json
    d = json.loads(line[-1]); print("   value", d["value"], "Msamples/s, ms/step", d["ms_per_step"], "steps", d["steps"])
