set -x
mkdir -p gpurun_out/r2i
timeout 600 python bench.py > gpurun_out/r2i/bench_default.json 2> gpurun_out/r2i/bench_default.err
timeout 300 python bench.py --precision bf16 --no-drift --cpu-seconds 0 > gpurun_out/r2i/bench_B16_bf16.json 2>/dev/null
timeout 300 python bench.py --ckpt ViT-L/14 --batch 256 --precision fp16 --no-drift --cpu-seconds 0 > gpurun_out/r2i/bench_L14_fp16.json 2>/dev/null
timeout 300 python bench.py --ckpt ViT-B/32 --precision fp16 --no-drift --cpu-seconds 0 > gpurun_out/r2i/bench_B32_fp16.json 2>/dev/null
timeout 300 python tools/e2e_probe.py > gpurun_out/r2i/e2e.txt 2>&1
tail -3 gpurun_out/r2i/e2e.txt
tail -c 1500 gpurun_out/r2i/bench_default.json
