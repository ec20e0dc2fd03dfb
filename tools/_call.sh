set -x
mkdir -p gpurun_out/r2b
for cfg in "20000 5000 1.5 0 fp16-exact" "20000 5000 1.5 1.0 fp32" "20000 5000 1.5 3.0 fp32" "20000 5000 0 2.0 fp32"; do
  set -- $cfg
  timeout 300 python -m mcm_amd.parity $cfg > gpurun_out/r2b/drift_$1_$3_$4_$5.json 2>> gpurun_out/r2b/drift.err
done
for shape in "100864 2304 768 0" "100864 3072 768 1" "100864 768 3072 2"; do
  timeout 120 ./tools/gemm_bench_trace $shape 5 0 0 3 2>&1 | grep -E "TRACE|step|BEST" >> gpurun_out/r2b/trace_v3.txt
  timeout 120 ./tools/gemm_bench_trace $shape 5 0 0 4 2>&1 | grep -E "TRACE|step|BEST" >> gpurun_out/r2b/trace_v4.txt
done
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r2b/pytest.txt 2>&1
tail -5 gpurun_out/r2b/pytest.txt
