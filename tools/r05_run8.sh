#!/bin/bash
# development (round 5, call 8): binning tail experiments (staggered view groups, batched slot requests, 44 KB histogram budget)
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05h
rm -rf "$O"; mkdir -p "$O"
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_raster.py -m gpu -x -q > "$O/pytest.log" 2>&1; tail -2 "$O/pytest.log" | cut -c1-200
for v in "stag -DSPF_VG_STAGGER=1" "tb -DSPF_TAIL_BATCH=1" "h44 -DSPF_HIST_BUDGET_KB=44"; do
  set -- $v
  SPF_LIB_DIR=_C_$1 SPF_HIPCC_EXTRA=$2 timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -q -k "direct_bins or full_batch or longest_first" > "$O/pytest_$1.log" 2>&1
  echo "$1: $(tail -1 $O/pytest_$1.log | cut -c1-150)"
done
for cfg in C2 C5 REF2V C3; do
  echo "== $cfg" | tee -a "$O/ab.txt"
  AB_ARGS="--config $cfg" tools/ab.sh _C _C_stag _C_tb _C_h44 2>&1 | tee -a "$O/ab.txt"
done
