#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_long4
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_long_sequences.py tests/test_gpu_class_plans.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
TOTAL=2e9 NSEQ=400 timeout 600 python scripts/dev/perf_long2.py > $O/long.txt 2>&1; cat $O/long.txt
BSK_SWEEP_OUTLIER_READS=1e8 timeout 900 python scripts/robustness_sweep.py 3e9 --only-outliers > $O/outliers.jsonl 2> $O/outliers.err
python - <<'PY'
import json
for l in open("gpurun_out/r6_long4/outliers.jsonl"):
    d = json.loads(l)
    if "bp" in str(d.get("case")) and d.get("gbases_per_s", 0) > 100: print(d["case"], d["gbases_per_s"], d.get("gbases_per_s_with_prepare"), d["kernel"][:110])
PY
