#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_long3
mkdir -p $O
BSK_SWEEP_OUTLIER_READS=1e8 timeout 900 python scripts/robustness_sweep.py 3e9 --only-outliers > $O/outliers.jsonl 2> $O/outliers.err
python - <<'PY'
import json
for l in open("gpurun_out/r6_long3/outliers.jsonl"):
    d = json.loads(l)
    if "bp" in str(d.get("case")) and d.get("gbases_per_s", 0) > 100: print(d["case"], d["gbases_per_s"], d.get("gbases_per_s_with_prepare"), d["kernel"][:110])
PY
