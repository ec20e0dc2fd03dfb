#!/bin/bash
# scratch driver (round 3, call 50): the head-major arm compiled out of the shipped library — kernel / model / arm tests, short bench
mkdir -p gpurun_out/r3c50
O=$PWD/gpurun_out/r3c50
( time timeout 2400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_qkv_layout.py tests/test_gpu_ln_tail.py tests/test_gpu_ln_fold.py tests/test_gpu_round2.py tests/test_gpu_c_abi.py -m gpu -x -q ) > $O/pytest.txt 2>&1
grep -E "passed|failed|^E |^real" $O/pytest.txt | head -5
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
for rep in 1 2; do
timeout 600 python bench.py --no-drift --cpu-seconds 0 --steps 40 > $O/b_$rep.json 2> $O/b_$rep.err || tail -3 $O/b_$rep.err
python - <<PY
import json
d=json.load(open("$O/b_$rep.json"))
print(round(d["value"]), d["ms_per_step"], round(d["sustained_images_per_sec"]), d["roofline"]["frac"], d["kernel_ms_per_step"])
PY
done
