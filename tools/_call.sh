set -x
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_round2.py -m gpu -q -k "large_sample or round_trip" 2>&1 | tail -8
bash tools/profile.sh r02_a
