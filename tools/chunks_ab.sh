for rep in 1 2; do
for mode in "1 --eager" "2 --eager" "1" "2"; do
  set -- $mode
  SPF_CHUNKS=$1 timeout 180 python bench.py --no-cpu-baseline --min-trials 15 $2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); st=d['stage_ms_per_step_warmup']
print('chunks $1 $2', d['value'], d['ms_per_step'], ' '.join(f'{k[:6]}={v*1e3:.1f}' for k, v in st.items()))"
done; done
