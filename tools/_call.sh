#!/bin/bash
# scratch driver (round 3, call 45): sliver-round split of the persistent GEMM (rows that fill whole rounds -> ping-pong
# kernel, the few row tiles left -> tile kernel) vs the unsplit library, at the checkpoints whose tile counts have slivers
mkdir -p gpurun_out/r3c45
O=$PWD/gpurun_out/r3c45
timeout 1800 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
one() {
  lib=$2; shift 2
  timeout 600 python tools/bench_with_lib.py mcm_amd/$lib --no-drift --cpu-seconds 0 --sustain-seconds 0 "$@" > $O/b_$tag.json 2> $O/b_$tag.err || tail -3 $O/b_$tag.err
  python - <<PY
import json
d=json.load(open("$O/b_$tag.json"))
print("$tag", round(d["value"]), round(d["ms_per_step"],3), d["kernel_ms_per_step"]["gemm"], round(d["roofline"]["frac"],4))
PY
}
for rep in 1 2; do
  for lib in nosplit split; do
    so=libmcm_hip.so; [ $lib = nosplit ] && so=libmcm_hip_nosplit.so
    tag=L14_b256_${lib}_$rep; one $tag $so --ckpt ViT-L/14 --batch 256 --steps 20
    tag=L14_b512_${lib}_$rep; one $tag $so --ckpt ViT-L/14 --batch 512 --steps 10
    tag=B32_b512_${lib}_$rep; one $tag $so --ckpt ViT-B/32 --batch 512 --steps 40
    tag=B16_b512_${lib}_$rep; one $tag $so --steps 30
  done
done 2>&1 | tee $O/bench.txt
