#!/bin/bash
# scratch driver (round 3, call 43): L2 prefetch of the residual rows (inside the epilogue / a compute phase early) vs none
mkdir -p gpurun_out/r3c43
O=$PWD/gpurun_out/r3c43
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_ln_fold.py tests/test_gpu_qkv_layout.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for gb in gemm_bench gemm_bench_early; do for shp in "768 3072 2" "768 768 2"; do
  echo $gb >> $O/gemm.txt; timeout 300 tools/$gb 100864 $shp 1500 0 0 3 0x20 2>&1 | grep -E "BEST|vs variant 0" >> $O/gemm.txt
done; done
cat $O/gemm.txt
one() {
  timeout 600 python tools/bench_with_lib.py mcm_amd/$2 --no-drift --cpu-seconds 0 --steps 40 > $O/b_$1.json 2> $O/b_$1.err || tail -3 $O/b_$1.err
  python - <<PY
import json
d=json.load(open("$O/b_$1.json"))
print("$1", round(d["value"]), d["ms_per_step"], round(d["sustained_images_per_sec"]), d["kernel_ms_per_step"], d["sustained"].get("sclk_mhz_mean"))
PY
}
for rep in 1 2 3; do one noprefetch_$rep libmcm_hip_scalar.so; one prefetch_$rep libmcm_hip.so; one early_$rep libmcm_hip_pfearly.so; done 2>&1 | tee $O/bench.txt
