#!/bin/bash
# development: A/B of a sort-launcher switch on one box.   tools/sort_ab.sh SPF_SORT_STATIC "C2 C5"
var=${1:-SPF_SORT_STATIC}; cfgs=${2:-"C2 C5"}
for rep in 1 2; do for on in 0 1; do for cfg in $cfgs; do
  if [ $on = 1 ]; then export $var=1; else unset $var; fi
  timeout 180 python bench.py --no-cpu-baseline --min-trials 15 --config $cfg 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); st=d['stage_ms_per_step_warmup']
print('$cfg $var=$on', d['value'], d['ms_per_step'], 'sort=%.1f' % (st['tile_sort']*1e3))"
done; done; done
