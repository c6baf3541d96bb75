#!/bin/bash
# development (round 5, first GPU call): tests, the default bench, rope heads A/B, REF10V kernel stats, two fuzz seeds
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05a
rm -rf "$O"; mkdir -p "$O"
export SPF_PARITY_REPORT=$O/parity_reports.jsonl
timeout 1500 python -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1; echo "pytest rc $?" >> "$O/pytest.log"
tail -5 "$O/pytest.log"
unset SPF_PARITY_REPORT
timeout 600 python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"; echo "bench rc $?"
for h in 1 2 4; do SPF_ROPE_HEADS=$h timeout 200 python bench.py --rope > "$O/rope_h$h.json" 2> "$O/rope_h$h.err"; done
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_REF10V" -o stats -- \
    python bench.py --config REF10V --no-cpu-baseline --no-secondary > "$O/stats_REF10V.log" 2>&1
find "$O/stats_REF10V" -name '*kernel_stats.csv' -exec cp {} "$O/kernel_stats_REF10V.csv" \;
rm -rf "$O/stats_REF10V"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_rope" -o stats -- \
    python bench.py --rope > "$O/stats_rope.log" 2>&1
find "$O/stats_rope" -name '*kernel_stats.csv' -exec cp {} "$O/kernel_stats_rope.csv" \;
rm -rf "$O/stats_rope"
for s in 170586 260130; do timeout 600 python tools/debug_seed.py $s --f32 > "$O/debug_$s.txt" 2>&1; done
find "$O" -name '*.log' -size +300k -delete
ls -la "$O"
