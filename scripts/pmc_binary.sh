#!/bin/bash
# dev: FETCH_SIZE / WRITE_SIZE per kernel of a stand-alone binary.  usage: scripts/pmc_binary.sh <binary> <tag>
BIN=$(realpath $1); TAG=$2
REPO=$(pwd); mkdir -p gpurun_out; OUT=$REPO/gpurun_out/pmcbin_$TAG.txt; : > $OUT
$BIN >> $OUT 2>&1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pb_$c
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pb_$c -- $BIN > /tmp/pb_$c.log 2>&1
  f=$(find /tmp/pb_$c -name '*counter_collection.csv' | head -1)
  python3 - "$f" $c >> $OUT <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2]:
        v = float(r["Counter_Value"]) * 1024 * (2 if sys.argv[2] == "FETCH_SIZE" else 1)
        print(f'{sys.argv[2]:10s} {r["Kernel_Name"][:60]:60s} {v/1e9:8.3f} GB')
PY
done
cat $OUT
