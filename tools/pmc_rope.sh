#!/bin/bash
# HBM counter passes for the RoPE kernel (run on the GPU box, repo root): FETCH_SIZE and WRITE_SIZE in separate runs of
# `bench.py --rope --eager` (the headline case only -- fp32 q at (48,256,16,64) -- launched one by one over the cold
# rotation of buffers) -> gpurun_out/pmc_summary_rope.json; copy to profiles/ to fill `roofline.traffic` of the rope line.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_rope
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- python bench.py --rope --eager > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- python bench.py --rope --eager > "$OUT/write.log" 2>&1
F=$(find "$OUT/fetch" -name '*counter_collection.csv' | head -1)
W=$(find "$OUT/write" -name '*counter_collection.csv' | head -1)
python tools/pmc_summary.py "$F" "$W" gpurun_out/pmc_summary_rope.json | tee gpurun_out/pmc_summary_rope.txt
rm -rf "$OUT/fetch" "$OUT/write"
