#!/bin/bash
# development: long campaign on seed ranges of its own (round 3, second campaign: 50000+ / 60000+ / 70000+)
mkdir -p gpurun_out
python tools/fuzz_campaign.py --wide --first 50000 --count 8000 --seconds ${1:-1500} --out gpurun_out/fuzz2_wide.jsonl
python tools/fuzz_campaign.py --first 60000 --count 6000 --seconds ${2:-900} --out gpurun_out/fuzz2_plain.jsonl
python tools/fuzz_campaign.py --large --first 70000 --count 1200 --seconds ${3:-900} --out gpurun_out/fuzz2_large.jsonl
grep -h '"fails": \["' gpurun_out/fuzz2_wide.jsonl gpurun_out/fuzz2_plain.jsonl gpurun_out/fuzz2_large.jsonl | grep -v '"inconclusive": true' | cut -c1-1800 | head -30
