#!/bin/bash
# scratch driver (round 4, call 1): split-weight arm — new tests, model tests, bench in both weight regimes
mkdir -p gpurun_out/r4c01
O=$PWD/gpurun_out/r4c01
timeout 900 python -m pytest tests/test_gpu_split_weights.py tests/test_gpu_c_abi.py tests/test_gpu_model.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
timeout 300 python bench.py --no-drift --cpu-seconds 0 > $O/bench_fp16exact.json 2> $O/bench_fp16exact.err; tail -c 600 $O/bench_fp16exact.json
timeout 300 python bench.py --no-drift --cpu-seconds 0 --weights-regime fp32 > $O/bench_fp32w_split.json 2> $O/bench_fp32w_split.err; tail -c 600 $O/bench_fp32w_split.json
