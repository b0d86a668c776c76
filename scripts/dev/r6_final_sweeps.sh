#!/bin/bash
# round 6, end of round: the off-bench evidence on the last library -> gpurun_out/r6_final/
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_final
mkdir -p $O
BSK_SWEEP_OUTLIER_READS=1e8 timeout 1200 python scripts/robustness_sweep.py 3e9 > $O/robustness.jsonl 2> $O/robustness.err
timeout 900 python scripts/sweep_distributions.py 3e9 > $O/distributions.jsonl 2> $O/distributions.err
TOTAL=2e9 NSEQ=400 timeout 300 python scripts/dev/perf_long2.py > $O/long.txt 2>&1
timeout 900 python scripts/dev/scan_plans.py 1.5e9 > $O/scan_plans.txt 2>&1
timeout 300 python scripts/dev/r6_heads.py > $O/heads.txt 2>&1
wc -l $O/*
