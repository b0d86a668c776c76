#!/bin/bash
# Runs ON THE GPU BOX: the A/B of the two minimizer mappings -- kernel time (rocprofv3 --kernel-trace --stats) and SQ counters
# for k_minimizer_fast<11> (one read per lane), k_minimizer_wpr<11> (one read per wavefront) and k_minimizer_seg<11> (12 waves per CU).
# -> gpurun_out/ab/
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/ab; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in fast wpr seg; do
  unset BSK_WPR BSK_SEG
  [ $v = wpr ] && export BSK_WPR=1
  [ $v = seg ] && export BSK_SEG=1
  rm -rf /tmp/ab_$v
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$v -- python $REPO/scripts/perf_quick.py 1e8 min 21 11 5 > $OUT/perf_$v.log 2>&1
  cp $(find /tmp/ab_$v -name '*kernel_stats.csv' | head -1) $OUT/${v}_kernel_stats.csv 2>/dev/null
  cd $REPO && bash scripts/pmc_sq.sh ab_$v "1e7 min 21 11 2" > /dev/null 2>&1; cp gpurun_out/sq_ab_$v.txt $OUT/${v}_sq_counters.txt; cd /tmp
done
grep -h "Gbases\|kernel ms" $OUT/perf_*.log
