cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d /root/repo/gpurun_out/prof_fold/fetch -o f -- python /root/repo/bench.py --steps 6 --warmup 2 --no-drift --cpu-seconds 0 --sustain-seconds 0 --no-profile > /root/repo/gpurun_out/prof_fold/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d /root/repo/gpurun_out/prof_fold/write -o w -- python /root/repo/bench.py --steps 6 --warmup 2 --no-drift --cpu-seconds 0 --sustain-seconds 0 --no-profile > /root/repo/gpurun_out/prof_fold/write.log 2>&1
