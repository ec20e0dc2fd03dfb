#!/bin/bash
# scratch driver (round 4, call 37): rocprofv3 records of the final tree (+ the ingest path's kernels)
bash tools/profile.sh r04_q > gpurun_out/prof_r04_q.log 2>&1; tail -8 gpurun_out/prof_r04_q.log
out=$PWD/gpurun_out/prof_r04_q_ingest; mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python $root/bench.py --steps 6 --warmup 2 --no-drift --cpu-seconds 0 --sustain-seconds 0 --no-profile --ingest host-raw --no-arms --no-live-traffic > $out/trace.log 2>&1
cd $root
tr=$(find $out/trace \( -name "*_results.db" -o -name "*kernel_stats.csv" \) | head -1)
case "$tr" in
  *.db) python tools/rocpd_summary.py $tr > $out/kernel_stats.txt ;;
  *.csv) cp $tr $out/kernel_stats.txt ;;
esac
mkdir -p gpurun_out/prof_r04_q/summaries; cp $out/kernel_stats.txt gpurun_out/prof_r04_q/summaries/r04_q_ingest_kernel_stats.txt
grep -i "resize\|kernel  " $out/kernel_stats.txt | head
