#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): arbitrary counter groups of one perf_quick.py workload, one rocprofv3 pass per group.
# usage: scripts/pmc_custom.sh <tag> "<perf_quick args>" "<group 1>" "<group 2>" ...   -> gpurun_out/pmc_<tag>.txt
set -u
TAG=$1; ARGS=$2; shift 2
REPO=$(pwd); mkdir -p "$REPO/gpurun_out"; OUT=$REPO/gpurun_out/pmc_$TAG.txt; : > "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for g in "$@"; do
  rm -rf /tmp/pc_$i
  timeout 300 rocprofv3 --pmc $g --output-format csv -d /tmp/pc_$i -- python $REPO/scripts/perf_quick.py $ARGS > /tmp/pc_$i.log 2>&1
  f=$(find /tmp/pc_$i -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then
    python - "$f" >> "$OUT" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
disp = collections.Counter(r["Kernel_Name"][:60] for r in rows)
for k, d in acc.items():
    if "bsk::k_" not in k: continue
    nd = disp[k] / max(len(d), 1)
    if max(d.values()) / nd < 1e4: continue
    print(k, "dispatches", int(nd), {c: v / nd for c, v in d.items()})
PY
  else
    echo "group $i ($g) failed: $(tail -2 /tmp/pc_$i.log)" >> "$OUT"
  fi
  i=$((i+1))
done
cat "$OUT"
