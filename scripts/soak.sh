#!/bin/bash
# soak + fuzz of the frame-stream scheduler (usage: soak.sh <tag> [first api-fuzz seed] [count]): random API sequences (bursts of one-frame submissions, pt_write_frame, image ring,
# input uploads, pool sizes) against a model built from the oracle, then 150 bench steps = 5 G jobs per stream pair: the streams' 31-bit job counters wrap (a stream restart inside
# the timed region), rings and image ring wrap many times; the line's parity block compares frames of the timed workload with the oracle.
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-soak}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
timeout -k 10 900 python3 scripts/api_fuzz.py ${2:-100001} ${3:-1500} 2>&1 | tail -4 | tee $O/api_fuzz.txt
timeout -k 10 900 python3 bench.py --steps 150 --warmup 2 > $O/soak.json 2> $O/soak.err; echo "soak rc=$?"
python3 -c "
import json
for l in open('$O/soak.json'):
    if l.startswith('{'):
        d=json.loads(l); print('soak', d['value'], 'Ms/s', d['ms_per_step'], 'ms/step', d['parity'])
" | tee $O/soak.txt
