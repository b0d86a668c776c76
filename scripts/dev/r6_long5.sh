#!/bin/bash
cd "$(dirname "$0")/../.."
for tp in 96 112 128; do
echo "== BSK_TILE_POS=$tp"
BSK_TILE_POS=$tp BSK_TIMING=1 TOTAL=2e9 NSEQ=400 timeout 600 python scripts/dev/perf_long2.py 2>&1 | grep -E "kernels|stitch|wall|sizing attempt" | head -12
done
