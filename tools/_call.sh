#!/bin/bash
# scratch driver (round 3, call 7): FPR95-count distribution over draws, the re-bounded tests
mkdir -p gpurun_out/r3c7
O=$PWD/gpurun_out/r3c7
timeout 900 python tools/drift_seeds.py fp16-exact 6 0 > $O/seeds_fp16exact.json 2> $O/seeds1.err; tail -6 $O/seeds1.err
timeout 900 python tools/drift_seeds.py fp32 4 0 > $O/seeds_fp32w.json 2> $O/seeds2.err; tail -4 $O/seeds2.err
timeout 900 python tools/drift_seeds.py fp16-exact 3 3.0 > $O/seeds_fp16exact_tile3.json 2> $O/seeds3.err; tail -3 $O/seeds3.err
( time timeout 2400 python -m pytest tests/test_gpu_configs.py tests/test_gpu_headline_parity.py -m gpu -q --durations=8 ) > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
