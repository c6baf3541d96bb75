#!/bin/bash
# development: regression campaign after the round-3 projection diet (frustum clamp decided by a multiplication, hardware
# sqrt / rcp for the cull disc, focal lengths per view through v_readlane) on seed ranges of its own: 80000+ / 90000+ / 100000+
mkdir -p gpurun_out
python tools/fuzz_campaign.py --wide --first 80000 --count 4000 --seconds ${1:-150} --out gpurun_out/fuzz4_wide.jsonl
python tools/fuzz_campaign.py --first 90000 --count 4000 --seconds ${2:-100} --out gpurun_out/fuzz4_plain.jsonl
python tools/fuzz_campaign.py --large --first 100000 --count 600 --seconds ${3:-120} --out gpurun_out/fuzz4_large.jsonl
grep -h '"fails": \["' gpurun_out/fuzz4_wide.jsonl gpurun_out/fuzz4_plain.jsonl gpurun_out/fuzz4_large.jsonl | grep -v '"inconclusive": true' | cut -c1-1500 | head -20
tail -qn1 gpurun_out/fuzz4_wide.jsonl gpurun_out/fuzz4_plain.jsonl gpurun_out/fuzz4_large.jsonl | cut -c1-1200
