#!/bin/bash
# scratch driver (round 4, call 11): 3-deep ingest pipes, ingest tests, metrics test, then the final records of the round
mkdir -p gpurun_out/r4c11
O=$PWD/gpurun_out/r4c11
timeout 900 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_metrics.py tests/test_gpu_round2.py -m gpu -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt | cut -c1-200
B="--no-drift --cpu-seconds 0 --no-arms --sustain-seconds 0"
for t in 16 16; do
MCM_PACK_THREADS=$t timeout 300 python bench.py $B --ingest host-raw,host-u8 > $O/bench_ingest$t.json 2> $O/bench_ingest$t.err
python - <<PY
import json
d=json.loads(open("$O/bench_ingest$t.json").read().strip().splitlines()[-1])
print("pack threads $t", round(d["value"]), {k:(round(v["images_per_sec"]), round(v["pcie_gb_per_sec"],1)) for k,v in d["ingest"].items()})
PY
done
# logic check of the N = 2 path on the one GPU (gloo) and smoke()
timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-drift --cpu-seconds 0 --sustain-seconds 0 > $O/bench_2ranks.json 2> $O/bench_2ranks.err; tail -c 400 $O/bench_2ranks.json
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
