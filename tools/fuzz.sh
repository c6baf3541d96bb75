#!/bin/bash
# development: the three fuzz-campaign families against the oracle on seed ranges of their own.
#   tools/fuzz.sh TAG FIRST_SEED [wide_seconds plain_seconds large_seconds]
# e.g. tools/fuzz.sh r4 140000 400 250 300  ->  gpurun_out/fuzz_r4_{wide,plain,large}.jsonl (seeds FIRST, FIRST+10000, FIRST+20000)
tag=${1:?tag}; first=${2:?first seed}
mkdir -p gpurun_out
python tools/fuzz_campaign.py --wide --first $first --count 8000 --seconds ${3:-400} --out gpurun_out/fuzz_${tag}_wide.jsonl
python tools/fuzz_campaign.py --first $((first + 10000)) --count 8000 --seconds ${4:-250} --out gpurun_out/fuzz_${tag}_plain.jsonl
python tools/fuzz_campaign.py --large --first $((first + 20000)) --count 1200 --seconds ${5:-300} --out gpurun_out/fuzz_${tag}_large.jsonl
grep -h '"fails": \["' gpurun_out/fuzz_${tag}_*.jsonl | grep -v '"inconclusive": true' | cut -c1-1500 | head -20
tail -qn1 gpurun_out/fuzz_${tag}_wide.jsonl gpurun_out/fuzz_${tag}_plain.jsonl gpurun_out/fuzz_${tag}_large.jsonl | cut -c1-600
