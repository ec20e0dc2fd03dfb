#!/bin/bash
# scratch driver (round 4, call 46): colour kernel with dword stores — tests + kernel trace of the JPEG ingest leg
mkdir -p gpurun_out/r4c46
timeout 900 python -m pytest tests/test_gpu_jpeg.py tests/test_gpu_ingest.py -x -q -m gpu 2>&1 | tail -3
out=$PWD/gpurun_out/prof_r04_v_jpeg; mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python $root/bench.py --steps 6 --warmup 2 --no-drift --cpu-seconds 0 --sustain-seconds 0 --no-profile --ingest host-jpeg --no-arms --no-live-traffic > $out/trace.log 2>&1
cd $root
tr=$(find $out/trace \( -name "*_results.db" -o -name "*kernel_stats.csv" \) | head -1)
case "$tr" in
  *.db) python tools/rocpd_summary.py $tr > $out/kernel_stats.txt ;;
  *.csv) cp $tr $out/kernel_stats.txt ;;
esac
grep -i "jpeg\|resize\|kernel  " $out/kernel_stats.txt | head
tail -1 $out/trace.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print({k: round(v.get('images_per_sec',-1)) for k,v in d['ingest'].items()})"
