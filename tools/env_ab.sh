#!/bin/bash
# development: A/B of an environment switch on one box.   tools/env_ab.sh VAR "valueA valueB" "C2 C3" [stage]
var=$1; vals=$2; cfgs=${3:-C2}; stage=${4:-bin_pairs}
for rep in 1 2; do for val in $vals; do for cfg in $cfgs; do
  env $var=$val timeout 180 python bench.py --no-cpu-baseline --min-trials 15 --config $cfg 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); st=d['stage_ms_per_step_warmup']
print('$cfg $var=$val', d['value'], d['ms_per_step'], '$stage=%.1f' % (st['$stage']*1e3))"
done; done; done
