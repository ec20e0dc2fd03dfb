#!/bin/bash
# scratch driver (round 4, call 5): refinement tests, grouped tile walk on the ping-pong kernel (arms text, variant 9) + fetch bytes
mkdir -p gpurun_out/r4c05
O=$PWD/gpurun_out/r4c05
timeout 1500 python -m pytest tests/test_gpu_headline_parity.py tests/test_gpu_configs.py tests/test_gpu_rccl.py tests/test_gpu_split_weights.py tests/test_gpu_kernels.py -m gpu -q -k "not test_linear and not pingpong_interior or full_size_variants or tile64" > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
B="--no-drift --cpu-seconds 0 --ingest none --no-arms --sustain-seconds 3"
for g in 0 4 6 3 0; do
  timeout 300 python bench.py $B --gemm-variant 9 --group-n $g > $O/bench_v9_gn$g.json 2> $O/bench_v9_gn$g.err
  python - <<PY
import json
d=json.loads(open("$O/bench_v9_gn$g.json").read().strip().splitlines()[-1])
print("variant 9 group_n $g", round(d["value"]), d["kernel_ms_per_step"], d.get("sustained",{}).get("sclk_mhz_mean"), d.get("sustained",{}).get("power_w_mean"))
PY
done
cd /tmp && export TMPDIR=/tmp
for g in 0 4; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/fetch_g$g -o f -- python /root/repo/bench.py --steps 4 --warmup 2 --no-drift --cpu-seconds 0 --sustain-seconds 0 --no-profile --ingest none --no-arms --gemm-variant 9 --group-n $g > $O/fetch_g$g.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/write_g$g -o w -- python /root/repo/bench.py --steps 4 --warmup 2 --no-drift --cpu-seconds 0 --sustain-seconds 0 --no-profile --ingest none --no-arms --gemm-variant 9 --group-n $g > $O/write_g$g.log 2>&1
  fe=$(find $O/fetch_g$g -name "*_results.db" -o -name "*counter_collection.csv" | head -1); wr=$(find $O/write_g$g -name "*_results.db" -o -name "*counter_collection.csv" | head -1)
  cd /root/repo; python tools/pmc_summary.py $fe $wr > $O/pmc_traffic_v9_gn$g.txt 2>&1; cd /tmp
  rm -rf $O/fetch_g$g $O/write_g$g
done
cd /root/repo; grep "gemm_pp" $O/pmc_traffic_v9_gn0.txt; grep "gemm_pp" $O/pmc_traffic_v9_gn4.txt
