#!/bin/bash
# development: same-box A/B of library builds.   tools/ab.sh _C _C_base [...]   (alternating, two rounds each)
# AB_ARGS passes extra bench.py arguments, e.g.  AB_ARGS="--config C5" tools/ab.sh _C _C_base
for rep in 1 2; do
  for lib in "$@"; do
    SPF_LIB_DIR=$lib timeout 180 python bench.py --no-cpu-baseline --no-secondary --min-trials 15 ${AB_ARGS:-} 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); st=d['stage_ms_per_step_warmup']
print('$lib', d['value'], d['ms_per_step'], 'dom', d['roofline']['launch_ms'], ' '.join(f'{k[:6]}={v*1e3:.1f}' for k, v in st.items()))"
  done
done
