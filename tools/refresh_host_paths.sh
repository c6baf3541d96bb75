set -u
OUT=$PWD/gpurun_out/refresh2; rm -rf "$OUT"; mkdir -p "$OUT"
tools/gpu_tests.sh 2>&1 | tail -3
cp gpurun_out/parity_reports.jsonl "$OUT/parity_reports.jsonl"
python bench.py > "$OUT/bench_C2.json" 2>> "$OUT/bench.err"
python bench.py --api module --no-cpu-baseline > "$OUT/bench_C2_api_module.json" 2>> "$OUT/bench.err"
SPF_NO_FAST=1 python bench.py --api module --no-cpu-baseline > "$OUT/bench_C2_api_module_python_step.json" 2>> "$OUT/bench.err"
SPF_PREPARE_STEPS=0 python bench.py --api module --no-cpu-baseline > "$OUT/bench_C2_api_module_general_path.json" 2>> "$OUT/bench.err"
python bench.py --eval-latency > "$OUT/bench_eval_1x3.json" 2>> "$OUT/bench.err"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/refresh2/bench_*.json")):
    d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"])
d = json.load(open("gpurun_out/refresh2/bench_C2.json")); print({k: v.get("value") for k, v in d["secondary"].items()})
PY
