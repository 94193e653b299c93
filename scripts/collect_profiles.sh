#!/bin/bash
# copies the summaries of a scripts/measure_all.sh pass (gpurun_out/<tag>) into profiles/ under the round's names:  collect_profiles.sh <tag> [round prefix, default r04]
O=gpurun_out/${1:?tag}; P=profiles; R=${2:-r04}
for c in C2 C3 C4 C5 C6; do [ -f $O/pmc_$c/summary.json ] && cp $O/pmc_$c/summary.json $P/pmc_$c.json && cp $O/pmc_$c/summary.txt $P/${R}_pmc_$c.txt; done
grep -h "^{" $O/bench_default_c3.json > $P/${R}_bench_default_c3.json
for f in bench_C1 bench_C2 bench_C4 bench_C5 bench_C6 bench_c3_one_stream bench_c3_untimed_kernels bench_torchrun_n1; do [ -f $O/$f.json ] && grep -h "^{" $O/$f.json > $P/${R}_$f.json; done
cp $O/rehearsals.txt $P/${R}_shard_rehearsals.txt
cp $O/trace_stats.txt $P/${R}_kernel_trace_stats_c3.txt
grep -o '"kernel_source_hash": "[0-9a-f]*"' $P/pmc_C3.json $P/${R}_bench_default_c3.json
