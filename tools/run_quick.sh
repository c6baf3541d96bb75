# development: GPU tests, then the C2 / REF2V / C5 bench lines (stage survey on stderr is dropped)
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
for cfg in C2 REF2V C5 C3; do
python bench.py --no-cpu-baseline --min-trials 15 --config $cfg 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); st=d['stage_ms_per_step_warmup']
print('$cfg', d['value'], d['ms_per_step'], 'dom', d['roofline']['kernel'][:22], d['roofline']['launch_ms'], d['roofline']['frac'], ' '.join(f'{k[:6]}={v*1e3:.1f}' for k, v in st.items()))"
done
