#!/bin/bash
# dev (GPU box): average shader clock during the sketch kernel = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the dispatch's duration,
# both from ONE rocprofv3 --pmc pass.  usage: scripts/dev/clock_probe.sh <tag> "<perf_quick args>"   (BSK_LIB / BSK_RING from the environment)
TAG=$1; ARGS=$2; REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cp_b
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/cp_b -- python $REPO/scripts/perf_quick.py $ARGS > /tmp/cp_b.log 2>&1
grep -E "Gbases" /tmp/cp_b.log
python - "$TAG" $(find /tmp/cp_b -name '*counter_collection.csv' | head -1) <<'PY'
import csv, sys, collections
tag, b = sys.argv[1:3]
rows = collections.defaultdict(list)
for r in csv.DictReader(open(b)):
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        rows[r["Kernel_Name"][:48]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), float(r["Counter_Value"])))
for k, v in rows.items():
    if "bsk::k_" not in k: continue
    v.sort()
    d, c = v[len(v) // 2]
    if d < 100000: continue
    print("%s %s: dispatch %.3f ms, GRBM_GUI_ACTIVE %.0f -> %.3f GHz" % (tag, k, d / 1e6, c, c / 8 / d))
PY
