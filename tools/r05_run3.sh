#!/bin/bash
# development (round 5, third GPU call): the 8 px build after the semantic-rect fix -- whole GPU suite; SQ counters of the
# composite kernels of both builds on C2; C3 with direct bins restored
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05c
rm -rf "$O"; mkdir -p "$O"
( export SPF_LIB_DIR=_C_w8 SPF_HIPCC_EXTRA=-DSPF_TILE=8
  timeout 1200 python -m pytest tests -m gpu -q > "$O/pytest_w8.log" 2>&1
  tail -30 "$O/pytest_w8.log" | cut -c1-200
  tools/pmc_passes.sh w8_C2 spf_render > "$O/pmc_w8.log" 2>&1
  cp gpurun_out/pmc_w8_C2.txt gpurun_out/sq_summary_w8_C2.json "$O/" )
tools/pmc_passes.sh t16_C2 spf_render > "$O/pmc_t16.log" 2>&1
cp gpurun_out/pmc_t16_C2.txt gpurun_out/sq_summary_t16_C2.json "$O/"
for cfg in C3 C2; do
  echo "== $cfg" | tee -a "$O/ab.txt"
  AB_ARGS="--config $cfg" tools/ab.sh _C _C_w8 2>&1 | tee -a "$O/ab.txt"
done
ls -la "$O"
