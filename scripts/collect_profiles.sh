#!/bin/bash
# copies the summaries of a scripts/measure_all.sh pass (gpurun_out/<tag>) into profiles/ under the round's names
O=gpurun_out/${1:?tag}; P=profiles
for c in C2 C3 C4 C5; do cp $O/pmc_$c/summary.json $P/pmc_$c.json; cp $O/pmc_$c/summary.txt $P/r03_m_pmc_$c.txt; done
grep -h "^{" $O/bench_default_c3.json > $P/r03_bench_default_c3.json
for f in bench_C2 bench_C4 bench_C5 bench_c3_one_stream bench_c3_untimed_kernels bench_torchrun_n1; do grep -h "^{" $O/$f.json > $P/r03_n_$f.json; done
cp $O/rehearsals.txt $P/r03_n_shard_rehearsals.txt
cp $O/trace_stats.txt $P/r03_n_kernel_trace_stats_c3.txt
grep -o '"kernel_source_hash": "[0-9a-f]*"' $P/pmc_C3.json $P/r03_bench_default_c3.json
