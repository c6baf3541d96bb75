#!/bin/bash
# development: L1 <-> L2 request counters of the projection kernels (is a kernel bound by line re-fetches rather than HBM?)
#   tools/l2_traffic.sh [bench args, e.g. --config REF2V]
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/l2work; rm -rf $OUT; mkdir -p $OUT
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -o p -- \
      python bench.py --eager --steps 3 --warmup 2 --min-trials 1 --min-seconds 0 --no-cpu-baseline "$@" > "$OUT/p$i.log" 2>&1
done
python tools/pmc_table.py $(find "$OUT" -name '*counter_collection.csv' | sort) --filter spf_project
