timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
bash tools/ab_chunks.sh
