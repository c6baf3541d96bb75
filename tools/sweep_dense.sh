#!/bin/bash
# sweep the sparse/dense tile threshold over footprint scales (run on the GPU box)
for s in ${SMULTS:-1 3 6 10}; do
  for thr in ${THRS:-12 20 28 40 60 100000}; do
    SPF_DENSE_AREA=$thr timeout 200 python bench.py --s-mult $s --steps 20 --warmup 4 --no-cpu-baseline --exact 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); st=d['stage_ms_per_step_warmup']; print('s_mult', $s, 'thr', $thr, 'ms', d['ms_per_step'], 'fwd', st['render_fwd'], 'bwd', st['render_bwd'])"
  done
done
