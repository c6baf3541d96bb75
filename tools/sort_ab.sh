for rep in 1 2; do for sep in 0 1; do for cfg in REF2V C3; do
  if [ $sep = 1 ]; then export SPF_SORT_SEPARATE=1; else unset SPF_SORT_SEPARATE; fi
  timeout 180 python bench.py --no-cpu-baseline --min-trials 15 --config $cfg 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); st=d['stage_ms_per_step_warmup']
print('$cfg separate=$sep', d['value'], d['ms_per_step'], 'sort=%.1f' % (st['tile_sort']*1e3))"
done; done; done
