#!/bin/bash
cd "$(dirname "$0")/../.."
TOTAL=2e8 NSEQ=40 timeout 120 python scripts/dev/perf_long2.py 2>&1 | tail -3
timeout 400 python -m pytest tests/test_gpu_long_sequences.py -x -q -m gpu -k "dense" 2>&1 | tail -15
TOTAL=2e9 NSEQ=400 timeout 200 python scripts/dev/perf_long2.py 2>&1 | tail -3
BSK_TIMING=1 TOTAL=2e9 NSEQ=400 timeout 200 python scripts/dev/perf_long2.py 2>&1 | grep -A8 "tile count" | head -20
