#!/bin/bash
# development: same-box A/B of the chunked two-lane chains (SPF_CHUNKS = 1 (off) / 2 / 4 / 8), two rounds.
# AB_ARGS passes extra bench.py arguments, e.g.  AB_ARGS="--config C5" tools/ab_chunks.sh
for rep in 1 2; do
  for c in ${CHUNKS:-1 2 4 8}; do
    SPF_CHUNKS=$c timeout 300 python bench.py --no-cpu-baseline --min-trials 15 ${AB_ARGS:-} 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); st=d['stage_ms_per_step_warmup']
print('chunks $c', d['value'], d['ms_per_step'], 'dom', d['roofline']['kernel'][:20], d['roofline']['launch_ms'], d['roofline']['frac'], ' '.join(f'{k[:6]}={v*1e3:.1f}' for k, v in st.items()))"
  done
done
