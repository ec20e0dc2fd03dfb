#!/bin/bash
# scratch driver (round 4, call 14): the default bench line (live PMC traffic, arms, ingest, parity) and the rocprofv3 passes of the shipped tree
mkdir -p gpurun_out/r4c14
O=$PWD/gpurun_out/r4c14
SECONDS=0
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench: $SECONDS s"; tail -c 600 $O/bench_default.json; tail -3 $O/bench_default.err
bash tools/profile.sh r04_j > $O/profile.log 2>&1; tail -5 $O/profile.log
