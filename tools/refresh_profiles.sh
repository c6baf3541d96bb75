#!/bin/bash
# Regenerate everything under profiles/ on a GPU box (run from the repo root).  Outputs go to gpurun_out/refresh/;
# copy them into profiles/ afterwards (tools/install_profiles.sh r03).   CONFIGS="C2 REF2V" limits the per-config part.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/refresh
rm -rf "$OUT"; mkdir -p "$OUT"
ERR="$OUT/bench.err"
for cfg in ${CONFIGS:-C2 REF2V C3 C5 REF10V}; do
  extra=""; [ "$cfg" != "C2" ] && extra="--no-cpu-baseline"
  # the bench line itself (graph replay, default batch of the config)
  # (C2 with default arguments is the driver's command: it carries the `secondary` object too)
  sec="--no-secondary"; [ "$cfg" = "C2" ] && sec=""
  python bench.py --config $cfg $extra $sec > "$OUT/bench_$cfg.json" 2>> "$ERR"
  # per-kernel durations of THE SAME command
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$cfg" -o stats -- \
      python bench.py --config $cfg --no-cpu-baseline --no-secondary > "$OUT/stats_$cfg.log" 2>&1
  find "$OUT/stats_$cfg" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats_$cfg.csv" \;
  # HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (eager launches: counters serialise the kernels anyway)
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_$cfg" -o fetch -- \
      python bench.py --config $cfg --eager --steps 5 --warmup 2 --min-trials 1 --min-seconds 0 --no-cpu-baseline > "$OUT/pmc_fetch_$cfg.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_$cfg" -o write -- \
      python bench.py --config $cfg --eager --steps 5 --warmup 2 --min-trials 1 --min-seconds 0 --no-cpu-baseline > "$OUT/pmc_write_$cfg.log" 2>&1
  F=$(find "$OUT/pmc_fetch_$cfg" -name '*counter_collection.csv' | head -1)
  W=$(find "$OUT/pmc_write_$cfg" -name '*counter_collection.csv' | head -1)
  python tools/pmc_summary.py "$F" "$W" "$OUT/pmc_summary_$cfg.json" > "$OUT/pmc_summary_$cfg.txt"
  # SQ counters (issue, LDS, waits), five passes
  PMC_ARGS="--config $cfg" tools/pmc_passes.sh refresh_$cfg spf_ > "$OUT/sq_passes_$cfg.log" 2>&1
  cp gpurun_out/pmc_refresh_$cfg.txt "$OUT/sq_counters_$cfg.txt"; cp gpurun_out/sq_summary_refresh_$cfg.json "$OUT/sq_summary_$cfg.json"
  rm -rf "$OUT/stats_$cfg" "$OUT/pmc_fetch_$cfg" "$OUT/pmc_write_$cfg"
done
if [ -z "${CONFIGS:-}" ]; then
  SPF_SH_BAND4=1 python bench.py --config REF2V --no-cpu-baseline > "$OUT/bench_REF2V_band4.json" 2>> "$ERR"
  python bench.py --streams 2 --no-cpu-baseline > "$OUT/bench_C2_streams2.json" 2>> "$ERR"
  python bench.py --eager --no-cpu-baseline --no-secondary > "$OUT/bench_C2_eager.json" 2>> "$ERR"
  python bench.py --scenes 64 --views 4 --no-cpu-baseline > "$OUT/bench_C2_64x4.json" 2>> "$ERR"
  SPF_DIRECT_BINS=0 python bench.py --no-cpu-baseline --no-secondary > "$OUT/bench_C2_classic_bins.json" 2>> "$ERR"
  python bench.py --eval-latency > "$OUT/bench_eval_1x3.json" 2>> "$ERR"
  python bench.py --api per-view --no-cpu-baseline --steps 5 --warmup 2 --min-trials 5 --min-seconds 0 > "$OUT/bench_perview.json" 2>> "$ERR"
  for v in "--sh-split" "--with-adapter" "--with-adapter --sh-split" "--raw-fused"; do
    n=$(echo "$v" | tr -d " -")
    SPF_SH_BAND4=0 python bench.py --config REF2V $v --no-cpu-baseline > "$OUT/bench_REF2V_$n.json" 2>> "$ERR"
  done
  for v in "--sh-split" "--with-adapter --sh-split" "--raw-fused"; do
    n=$(echo "$v" | tr -d " -")
    SPF_SH_BAND4=0 python bench.py --config REF10V $v --no-cpu-baseline > "$OUT/bench_REF10V_$n.json" 2>> "$ERR"
  done
  python bench.py --api module --no-cpu-baseline > "$OUT/bench_C2_api_module.json" 2>> "$ERR"
  SPF_PREPARE_STEPS=0 python bench.py --api module --no-cpu-baseline > "$OUT/bench_C2_api_module_general_path.json" 2>> "$ERR"
  python bench.py --s-mult 10 --no-cpu-baseline > "$OUT/bench_C2_stress.json" 2>> "$ERR"
  SPF_SH_BAND4=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_REF2V_rawfused" -o stats -- \
      python bench.py --config REF2V --raw-fused --no-cpu-baseline --no-secondary > "$OUT/stats_REF2V_rawfused.log" 2>&1
  find "$OUT/stats_REF2V_rawfused" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats_REF2V_rawfused.csv" \;
  rm -rf "$OUT/stats_REF2V_rawfused"
  SPF_SH_BAND4=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_REF2V_split" -o stats -- \
      python bench.py --config REF2V --sh-split --no-cpu-baseline --no-secondary > "$OUT/stats_REF2V_split.log" 2>&1
  find "$OUT/stats_REF2V_split" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats_REF2V_split.csv" \;
  rm -rf "$OUT/stats_REF2V_split"
  python bench.py --rope > "$OUT/rope_bench.json" 2>> "$ERR"
  tools/pmc_rope.sh > "$OUT/pmc_rope.log" 2>&1; cp gpurun_out/pmc_summary_rope.json "$OUT/pmc_summary_rope.json"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_rope" -o stats -- python bench.py --rope > "$OUT/stats_rope.log" 2>&1
  find "$OUT/stats_rope" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats_rope.csv" \;
  rm -rf "$OUT/stats_rope"
fi
find "$OUT" -name '*.log' -size +200k -delete
ls -la "$OUT"
