#!/bin/bash
# development (round 5, call 6): lists backward with longer rounds for the few-tiles regime
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05f
rm -rf "$O"; mkdir -p "$O"
for cfg in REF10V C3 C5; do
  echo "== $cfg" | tee -a "$O/ab.txt"
  AB_ARGS="--config $cfg" tools/ab.sh _C _C_r224 _C_r256 2>&1 | tee -a "$O/ab.txt"
done
