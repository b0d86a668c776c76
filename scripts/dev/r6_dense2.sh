#!/bin/bash
cd "$(dirname "$0")/../.."
for tk in 146 402 914 274 22; do
echo "== tk $tk"
BSK_PFT_TK=$tk BSK_TIMING=1 TOTAL=2e9 NSEQ=400 ONLY=minimizer timeout 100 python scripts/dev/perf_long2.py 2>&1 | grep -E "kernels|wall|again" | tail -3
done
