#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_long1
mkdir -p $O
TOTAL=2e9 NSEQ=400 BSK_TIMING=1 timeout 600 python scripts/dev/perf_long2.py > $O/long_timing.txt 2>&1
TOTAL=2e9 NSEQ=400 timeout 600 python scripts/dev/perf_long2.py > $O/long.txt 2>&1
cat $O/long.txt; tail -60 $O/long_timing.txt
