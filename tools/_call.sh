set -x
mkdir -p gpurun_out/r2e
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2e/pytest.txt 2>&1
tail -8 gpurun_out/r2e/pytest.txt
G=gpurun_out/r2e/gemm_sched.txt
for shape in "100864 2304 768 0" "100864 3072 768 1" "100864 768 3072 2" "100864 768 768 2"; do
  echo "== shape $shape" >> $G
  timeout 200 ./tools/gemm_bench $shape 10 0 0 3 0 104 2>&1 | grep -E "BEST|differing" >> $G
done
cat $G
timeout 300 python bench.py --steps 40 --warmup 5 --no-drift --cpu-seconds 0 --precision bf16 > gpurun_out/r2e/bench_bf16.json 2> gpurun_out/r2e/bench_bf16.err
timeout 300 python bench.py --steps 40 --warmup 5 --no-drift --cpu-seconds 0 --precision fp16 > gpurun_out/r2e/bench_fp16.json 2> gpurun_out/r2e/bench_fp16.err
timeout 400 python -m mcm_amd.parity 50000 10000 1.5 0 fp16-exact > gpurun_out/r2e/drift_50k_fp16exact_txt32.json 2> gpurun_out/r2e/drift.err
timeout 400 python -m mcm_amd.parity 50000 10000 1.5 0 fp32 > gpurun_out/r2e/drift_50k_fp32w_txt32.json 2>> gpurun_out/r2e/drift.err
tail -c 1500 gpurun_out/r2e/bench_fp16.json
