#!/bin/bash
# development (round 5, call 5): would coalesced SH reads pay?  (timing ablation _C_shco: wrong colours, same bytes)
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05e
rm -rf "$O"; mkdir -p "$O"
timeout 900 python -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1; tail -2 "$O/pytest.log" | cut -c1-200
echo "== REF2V band4" | tee -a "$O/ab.txt"
SPF_SH_BAND4=1 AB_ARGS="--config REF2V" tools/ab.sh _C _C_shco 2>&1 | tee -a "$O/ab.txt"
for cfg in REF2V C5 REF10V; do
  echo "== $cfg" | tee -a "$O/ab.txt"
  AB_ARGS="--config $cfg" tools/ab.sh _C _C_shco 2>&1 | tee -a "$O/ab.txt"
done
python bench.py --eval-latency > "$O/eval.json" 2> "$O/eval.err"; python -c "
import json; d=json.load(open('$O/eval.json')); print(d['latency_ms'])"
