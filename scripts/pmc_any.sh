#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): FETCH_SIZE / WRITE_SIZE passes of the bsk kernels of ANY python command (pmc_traffic.sh is the
# perf_quick.py special case).  usage: scripts/pmc_any.sh <tag> <script.py> [args...]   -> gpurun_out/traffic_<tag>.txt
set -u
TAG=$1; shift
REPO=$(pwd); mkdir -p "$REPO/gpurun_out"; OUT=$REPO/gpurun_out/traffic_$TAG.txt; : > "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tr_$c
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/tr_$c -- python $REPO/"$@" > /tmp/tr_$c.log 2>&1
  f=$(find /tmp/tr_$c -name '*counter_collection.csv' | head -1)
  python - "$f" $c >> "$OUT" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if r["Counter_Name"] == sys.argv[2]: acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "bsk::k_" in k and max(v) > 1e4:
        m = max(v) * 1024 * (2 if sys.argv[2] == "FETCH_SIZE" else 1)   # KiB; gfx950 counts 128-byte reads as 64
        print(f"{sys.argv[2]} {k}: {m/1e9:.3f} GB per launch (max of {len(v)} dispatches)")
PY
done
cat "$OUT"
