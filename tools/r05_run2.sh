#!/bin/bash
# development (round 5, second GPU call): the 8 px / one-wave-per-tile build (_C_w8) -- parity tests, then A/B against _C
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05b
rm -rf "$O"; mkdir -p "$O"
( export SPF_LIB_DIR=_C_w8 SPF_HIPCC_EXTRA=-DSPF_TILE=8
  timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_known_answers.py tests/test_gpu_raster_more.py -m gpu -q -x > "$O/pytest_w8_a.log" 2>&1
  tail -15 "$O/pytest_w8_a.log"
  timeout 900 python -m pytest tests/test_gpu_raster_fuzz.py tests/test_gpu_configs.py -m gpu -q > "$O/pytest_w8_b.log" 2>&1
  tail -15 "$O/pytest_w8_b.log" )
timeout 300 python -m pytest tests/test_gpu_eval_graphs.py tests/test_gpu_rope.py -m gpu -q > "$O/pytest_misc.log" 2>&1; tail -3 "$O/pytest_misc.log"
for cfg in C2 C3 C5 REF10V REF2V; do
  echo "== $cfg" | tee -a "$O/ab.txt"
  AB_ARGS="--config $cfg" tools/ab.sh _C _C_w8 2>&1 | tee -a "$O/ab.txt"
done
python bench.py --eval-latency > "$O/eval.json" 2> "$O/eval.err"; tail -c 700 "$O/eval.json"
ls -la "$O"
