#!/bin/bash
# scratch driver (round 4, call 24): resize kernel LDS form v2 — parity tests, A/B timing, ingest probe
mkdir -p gpurun_out/r4c24
O=$PWD/gpurun_out/r4c24
timeout 300 python tools/_dbg_resize.py 2>&1 | grep -v amdgpu.ids | grep -c "bad 0.0 " 
timeout 900 python -m pytest tests/test_gpu_preprocess.py tests/test_gpu_ingest.py tests/test_gpu_round2.py -x -q -m gpu > $O/pytest_resize.txt 2>&1; tail -3 $O/pytest_resize.txt
timeout 300 python tools/resize_bench.py > $O/resize_auto.jsonl 2> $O/resize_auto.err; cat $O/resize_auto.jsonl
timeout 300 python tools/resize_bench.py --fused-only > $O/resize_fused.jsonl 2> $O/resize_fused.err; cat $O/resize_fused.jsonl
timeout 600 python tools/ingest_probe.py > $O/ingest_probe.jsonl 2> $O/ingest_probe.err; grep -v '"pack"' $O/ingest_probe.jsonl; tail -3 $O/ingest_probe.err
