#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 600 python -m pytest tests/test_gpu_long_sequences.py tests/test_gpu_class_plans.py -x -q -m gpu 2>&1 | tail -4
TOTAL=2e9 NSEQ=400 timeout 200 python scripts/dev/perf_long2.py 2>&1 | tail -3
BSK_TILE_DENSE=1 TOTAL=2e9 NSEQ=400 ONLY=minimizer timeout 200 python scripts/dev/perf_long2.py 2>&1 | tail -2
BSK_TIMING=1 TOTAL=2e9 NSEQ=400 ONLY=minimizer timeout 200 python scripts/dev/perf_long2.py 2>&1 | grep "tiled" | tail -8
