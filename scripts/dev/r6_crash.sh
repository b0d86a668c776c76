#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 600 python scripts/dev/fuzz_range.py 21002300 21002760 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_long_sequences.py tests/test_gpu_class_plans.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3
timeout 1500 python scripts/fuzz_campaign.py 21000000 12000 2>&1 | tail -3
