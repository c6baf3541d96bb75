#!/bin/bash
# development: SQ counter passes for the bench step (run on the GPU box); output: gpurun_out/pmc_<tag>.txt and
# gpurun_out/sq_summary_<tag>.json       tools/pmc_passes.sh <tag> [filter]     (honours SPF_LIB_DIR / SPF_HIPCC_EXTRA;
# PMC_ARGS = extra bench.py arguments, e.g. "--config REF2V")
# (counters in their own runs, --kernel-trace only: no sys/hip/hsa trace domains next to --pmc)
tag=$1; flt=${2:-spf_render}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmcwork_$tag
mkdir -p "$OUT"
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS" \
           "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -o p -- \
      python bench.py --eager --steps 3 --warmup 2 --min-trials 1 --min-seconds 0 --no-cpu-baseline ${PMC_ARGS:-} > "$OUT/p$i.log" 2>&1
done
CSVS=$(find "$OUT" -name '*counter_collection.csv' | sort)
python tools/pmc_table.py $CSVS --filter "$flt" > gpurun_out/pmc_$tag.txt
python tools/sq_summary.py $CSVS gpurun_out/sq_summary_$tag.json > gpurun_out/sq_summary_$tag.txt
rm -rf "$OUT"
tail -40 gpurun_out/sq_summary_$tag.txt
