#!/bin/bash
# scratch driver (round 4, call 13): hipGraph replay of the step at small batches; new tests
mkdir -p gpurun_out/r4c13
O=$PWD/gpurun_out/r4c13
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_headline_parity.py -m gpu -q -k "graph or score_kind" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt | cut -c1-200
B="--no-drift --cpu-seconds 0 --no-arms --sustain-seconds 0 --ingest none --steps 60 --warmup 5"
for b in 1 4 8 16 32 64 128 512; do
for g in "" "--graph"; do
timeout 300 python bench.py $B --batch $b $g > $O/bench_b${b}${g}.json 2> $O/bench_b${b}${g}.err
python - <<PY
import json
d=json.loads(open("$O/bench_b${b}${g}.json").read().strip().splitlines()[-1])
print("batch $b $g", round(d["value"]), round(d["ms_per_step"],3))
PY
done
done
