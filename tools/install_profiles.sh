#!/bin/bash
# copy the outputs of tools/refresh_profiles.sh into profiles/ under a round prefix:   tools/install_profiles.sh r03
p=${1:?round prefix}
src=gpurun_out/refresh
for f in $src/bench_*.json $src/kernel_stats_*.csv $src/sq_counters_*.txt $src/rope_bench.json; do
  [ -f "$f" ] && cp "$f" profiles/${p}_$(basename "$f")
done
# what bench.py reads (roofline.traffic / roofline.valu): un-prefixed, one per config
for f in $src/pmc_summary_*.json $src/sq_summary_*.json; do [ -f "$f" ] && cp "$f" profiles/$(basename "$f"); done
ls profiles | grep "^$p" | head -60
