#!/bin/bash
# development: the three campaign families on seed ranges of their own (round 3: 20000+, never drawn before)
mkdir -p gpurun_out
python tools/fuzz_campaign.py --wide --first 20000 --count 2000 --seconds ${1:-400} --out gpurun_out/fuzz_wide.jsonl
python tools/fuzz_campaign.py --first 30000 --count 2000 --seconds ${2:-300} --out gpurun_out/fuzz_plain.jsonl
python tools/fuzz_campaign.py --large --first 40000 --count 400 --seconds ${3:-300} --out gpurun_out/fuzz_large.jsonl
grep -h '"fails": \["' gpurun_out/fuzz_wide.jsonl gpurun_out/fuzz_plain.jsonl gpurun_out/fuzz_large.jsonl | grep -v '"inconclusive": true' | cut -c1-1500 | head -40
