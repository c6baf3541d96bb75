run() { python bench.py --config C2 --s-mult 10 --no-cpu-baseline --no-secondary --min-trials 10 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); st=d['stage_ms_per_step_warmup']
print('$1', d['value'], d['ms_per_step'], ' '.join(f'{k[:10]}={v*1e3:.1f}' for k, v in st.items()))"; }
run default
SPF_DENSE_AREA=100000 run lists192
SPF_DENSE_AREA=100000 SPF_BWD_ROUNDS=224 run lists224
SPF_DENSE_AREA=100000 SPF_BWD_ROUNDS=256 run lists256
SPF_BWD_ROUNDS=256 run rows256
SPF_BWD_ROUNDS=224 run rows224
