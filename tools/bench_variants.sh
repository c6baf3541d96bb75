#!/bin/bash
# development: compare variant builds (SPF_LIB_DIR=_C_<name>, built with SPF_HIPCC_EXTRA=<flags>) on the GPU box
#   tools/bench_variants.sh "v0:-DSPF_CVAR=0" "v1:-DSPF_CVAR=1" ...
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  for rep in 1 2; do
    SPF_LIB_DIR=_C_$name SPF_HIPCC_EXTRA="$flags" python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); st=d['stage_ms_per_step_warmup']
print('$name', d['value'], d['ms_per_step'], ' '.join(f'{k}={v:.4f}' for k, v in st.items()))"
  done
done
