#!/bin/bash
# scratch driver (round 4, call 4): whole GPU suite on the refactored tree (gemm.hip / gemm_arms.hpp), N-split A/B + fetch bytes
mkdir -p gpurun_out/r4c04
O=$PWD/gpurun_out/r4c04
timeout 2400 python -m pytest tests -m gpu -q --durations=15 > $O/pytest.txt 2>&1; tail -30 $O/pytest.txt
B="--no-drift --cpu-seconds 0 --ingest none --no-arms --sustain-seconds 3"
for n in 1 2 3 4 1; do
  timeout 300 python bench.py $B --attn-variant 1 --nsplit $n > $O/bench_nsplit$n.json 2> $O/bench_nsplit$n.err
  python - <<PY
import json
d=json.loads(open("$O/bench_nsplit$n.json").read().strip().splitlines()[-1])
print("nsplit $n", round(d["value"]), d["kernel_ms_per_step"], d.get("sustained",{}).get("sclk_mhz_mean"), d.get("sustained",{}).get("power_w_mean"))
PY
done
cd /tmp && export TMPDIR=/tmp
for n in 1 3; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/fetch_n$n -o f -- python /root/repo/bench.py --steps 4 --warmup 2 --no-drift --cpu-seconds 0 --sustain-seconds 0 --no-profile --ingest none --no-arms --attn-variant 1 --nsplit $n > $O/fetch_n$n.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/write_n$n -o w -- python /root/repo/bench.py --steps 4 --warmup 2 --no-drift --cpu-seconds 0 --sustain-seconds 0 --no-profile --ingest none --no-arms --attn-variant 1 --nsplit $n > $O/write_n$n.log 2>&1
  fe=$(find $O/fetch_n$n -name "*_results.db" -o -name "*counter_collection.csv" | head -1); wr=$(find $O/write_n$n -name "*_results.db" -o -name "*counter_collection.csv" | head -1)
  cd /root/repo; python tools/pmc_summary.py $fe $wr > $O/pmc_traffic_nsplit$n.txt 2>&1; cd /tmp
  rm -rf $O/fetch_n$n $O/write_n$n
done
cd /root/repo; head -30 $O/pmc_traffic_nsplit1.txt; head -30 $O/pmc_traffic_nsplit3.txt
