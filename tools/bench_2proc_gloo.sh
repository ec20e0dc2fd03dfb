#!/bin/bash
# Exercises bench.py's N>1 code path (rank sharding, timed all-gather, max-over-ranks) on a ONE-GPU
# box: two processes share cuda:0 and talk over gloo instead of RCCL.  A logic check for the
# multi-process path, not a performance number.
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 WORLD_SIZE=2 LOCAL_RANK=0
for r in 0 1; do
  RANK=$r python - "$@" <<'PY' &
import os, sys
import torch.distributed as dist
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=2)
sys.argv = ["bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1", "--batch", "128", "--cpu-seconds", "0"] + sys.argv[1:]
import bench
bench.main()
PY
done
wait
