#!/bin/bash
# development (round 5, call 7): round shapes by tile count + prefetching instantiations, env-pinned A/B
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05g
rm -rf "$O"; mkdir -p "$O"
timeout 1200 python -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1; tail -3 "$O/pytest.log" | cut -c1-200
for cfg in REF10V C3 C2 C5 REF2V; do
  echo "== $cfg" | tee -a "$O/ab.txt"
  for rep in 1 2; do
    for v in "0 0" "1 0" "0 1" "1 1"; do
      set -- $v
      echo -n "fwdpf=$1 bwdpf=$2 " | tee -a "$O/ab.txt"
      SPF_FWD_PREFETCH=$1 SPF_BWD_PREFETCH=$2 AB_ARGS="--config $cfg" tools/ab.sh _C 2>&1 | head -1 | tee -a "$O/ab.txt"
    done
  done
done
