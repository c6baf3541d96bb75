#!/bin/bash
# development (round 5, fourth GPU call): contracted direction gradient (ABI 5), register sort for 2049..4096, SH burst A/B
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05d
rm -rf "$O"; mkdir -p "$O"
export SPF_PARITY_REPORT=$O/parity_reports.jsonl
timeout 1500 python -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1; tail -4 "$O/pytest.log" | cut -c1-200
unset SPF_PARITY_REPORT
for cfg in REF2V REF10V C5; do
  echo "== $cfg" | tee -a "$O/ab.txt"
  AB_ARGS="--config $cfg" tools/ab.sh _C _C_burst 2>&1 | tee -a "$O/ab.txt"
done
echo "== REF2V band4" | tee -a "$O/ab.txt"
SPF_SH_BAND4=1 AB_ARGS="--config REF2V" tools/ab.sh _C _C_burst 2>&1 | tee -a "$O/ab.txt"
for cfg in C2 C3; do
  echo "== $cfg" | tee -a "$O/ab.txt"
  AB_ARGS="--config $cfg" tools/ab.sh _C 2>&1 | tee -a "$O/ab.txt"
done
echo "== REF10V, LDS sort for 2049..4096 (round 4)" | tee -a "$O/ab.txt"
SPF_SORT_LDS_2K=1 AB_ARGS="--config REF10V" tools/ab.sh _C 2>&1 | tee -a "$O/ab.txt"
python bench.py --eval-latency > "$O/eval.json" 2> "$O/eval.err"; python -c "
import json; d=json.load(open('$O/eval.json')); print(d['latency_ms'])"
ls -la "$O"
