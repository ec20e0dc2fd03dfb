set -x
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6
timeout 300 python bench.py --no-drift --cpu-seconds 0 2>/dev/null | tail -c 700
