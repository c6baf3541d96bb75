set -x
export SPF_PARITY_REPORT=$PWD/gpurun_out/parity_reports.jsonl
rm -f $SPF_PARITY_REPORT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
