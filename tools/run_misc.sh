timeout 1500 python -m pytest tests/test_gpu_rope.py tests/test_gpu_multirank.py -m gpu -x -q 2>&1 | tail -6
python tools/bench_rope.py 2>&1 | grep -v amdgpu
python bench.py --api per-view --no-cpu-baseline --steps 5 --warmup 2 --min-trials 5 --min-seconds 0 > gpurun_out/bench_perview.json 2> gpurun_out/bench_perview.err; tail -3 gpurun_out/bench_perview.err; python -c "
import json; d=json.load(open('gpurun_out/bench_perview.json')); print('per-view', d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['stage_ms_per_step_warmup'])"
