#!/bin/bash
# development: campaign on the FINAL kernels of round 3 (merged / pair sort launches, pair slot at staging, camera half-angle,
# unit-gradient loss) on seed ranges of its own: 110000+ / 120000+ / 130000+
mkdir -p gpurun_out
python tools/fuzz_campaign.py --wide --first 110000 --count 8000 --seconds ${1:-400} --out gpurun_out/fuzz5_wide.jsonl
python tools/fuzz_campaign.py --first 120000 --count 8000 --seconds ${2:-250} --out gpurun_out/fuzz5_plain.jsonl
python tools/fuzz_campaign.py --large --first 130000 --count 1200 --seconds ${3:-300} --out gpurun_out/fuzz5_large.jsonl
grep -h '"fails": \["' gpurun_out/fuzz5_wide.jsonl gpurun_out/fuzz5_plain.jsonl gpurun_out/fuzz5_large.jsonl | grep -v '"inconclusive": true' | cut -c1-1500 | head -20
tail -qn1 gpurun_out/fuzz5_wide.jsonl gpurun_out/fuzz5_plain.jsonl gpurun_out/fuzz5_large.jsonl | cut -c1-400
