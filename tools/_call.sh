set -x
mkdir -p gpurun_out/r2f
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r2f/pytest.txt 2>&1
tail -30 gpurun_out/r2f/pytest.txt
