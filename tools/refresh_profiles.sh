#!/bin/bash
# Regenerate everything under profiles/ on a GPU box (run from the repo root).  Outputs go to gpurun_out/refresh/;
# copy them into profiles/ afterwards (see profiles/README.md for the names).
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/refresh
mkdir -p "$OUT"
python bench.py > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"
python bench.py --config C3 --no-cpu-baseline > "$OUT/bench_c3.json" 2>> "$OUT/bench_c2.err"
python bench.py --config C5 --no-cpu-baseline > "$OUT/bench_c5.json" 2>> "$OUT/bench_c2.err"
python bench.py --config REF2V --no-cpu-baseline > "$OUT/bench_ref2v.json" 2>> "$OUT/bench_c2.err"
SPF_SH_BAND4=1 python bench.py --config REF2V --no-cpu-baseline > "$OUT/bench_ref2v_band4.json" 2>> "$OUT/bench_c2.err"
python bench.py --streams 2 --no-cpu-baseline > "$OUT/bench_c2_streams2.json" 2>> "$OUT/bench_c2.err"
python bench.py --eager --no-cpu-baseline > "$OUT/bench_c2_eager.json" 2>> "$OUT/bench_c2.err"
python tools/bench_rope.py > "$OUT/rope_bench.json" 2>> "$OUT/bench_c2.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- \
    python bench.py --steps 50 --warmup 10 --no-cpu-baseline > "$OUT/stats.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o fetch -- \
    python bench.py --eager --steps 5 --warmup 2 --min-trials 1 --min-seconds 0 --no-cpu-baseline > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o write -- \
    python bench.py --eager --steps 5 --warmup 2 --min-trials 1 --min-seconds 0 --no-cpu-baseline > "$OUT/pmc_write.log" 2>&1
F=$(find "$OUT/pmc_fetch" -name '*counter_collection.csv' | head -1)
W=$(find "$OUT/pmc_write" -name '*counter_collection.csv' | head -1)
python tools/pmc_summary.py "$F" "$W" "$OUT/pmc_summary.json"
find "$OUT/stats" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats_bench_c2.csv" \;
tools/pmc_passes.sh refresh spf_ > "$OUT/sq_passes.log" 2>&1
cp gpurun_out/pmc_refresh.txt "$OUT/sq_counters.txt"; cp gpurun_out/sq_summary_refresh.json "$OUT/sq_summary.json"
# keep the merge-back small
find "$OUT" -name '*kernel_trace.csv' -delete
find "$OUT" -name '*counter_collection.csv' -delete
find "$OUT" -name '*agent_info.csv' -delete
ls -la "$OUT"
