#!/bin/bash
# profile.sh <tag> [bench args...] — the rocprofv3 passes behind profiles/<tag>_*:
#   1. --kernel-trace --stats                          -> <tag>_kernel_stats.csv (per-kernel time)
#   2. --pmc SQ_* + GRBM_GUI_ACTIVE  (own run)          -> <tag>_pmc_sq.txt  (MFMA busy %, LDS bank-conflict %)
#   3. --pmc FETCH_SIZE ...          (own run)          -> <tag>_pmc_traffic.txt
#   4. --pmc WRITE_SIZE ...          (own run)             + <tag>_traffic.json (HBM bytes per launch)
# PMC passes never share a run with tracing domains other than --kernel-trace (gpurun refuses that).
# Run on the GPU box from the repo root; writes gpurun_out/prof_<tag>/ and copies the summaries to profiles/.
set -u
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out $root/profiles
bench="python $root/bench.py --quick --steps 6 --warmup 2 --no-profile $*"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o t -- $bench > $out/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $out/sq -o s -- $bench > $out/sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $out/fetch -o f -- $bench > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $out/write -o w -- $bench > $out/write.log 2>&1
cd $root
# rocprofv3 7.2 writes rocpd SQLite databases by default (older builds: CSV); the summary tools read either
first() { find $1 \( -name "*_results.db" -o -name "$2" \) | head -1; }
tr=$(first $out/trace "*kernel_stats.csv")
sq=$(first $out/sq "*counter_collection.csv")
fe=$(first $out/fetch "*counter_collection.csv")
wr=$(first $out/write "*counter_collection.csv")
case "$tr" in
  *.db) python tools/rocpd_summary.py $tr > profiles/${tag}_kernel_stats.txt ;;
  *.csv) cp $tr profiles/${tag}_kernel_stats.csv ;;
esac
[ -n "$sq" ] && python tools/pmc_summary.py $sq > profiles/${tag}_pmc_sq.txt
[ -n "$fe" ] && [ -n "$wr" ] && python tools/pmc_summary.py $fe $wr > profiles/${tag}_pmc_traffic.txt
[ -n "$fe" ] && [ -n "$wr" ] && python tools/traffic_json.py $fe $wr profiles/${tag}_traffic.json > /dev/null
# profiles/ on the GPU box does not travel back (only gpurun_out/ is merged): keep a copy of the summaries next to the raw data
mkdir -p $out/summaries && cp profiles/${tag}_* $out/summaries/ 2>/dev/null
ls -la profiles/${tag}_* ; tail -3 $out/trace.log
