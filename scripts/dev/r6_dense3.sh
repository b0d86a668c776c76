#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
R=$PWD
rm -rf /tmp/prof_x
(cd /tmp && TOTAL=2e9 NSEQ=400 NODIGEST=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -- python $R/scripts/dev/perf_long2.py 2>&1 | grep wall)
python - <<PY
import csv,glob
f=glob.glob('/tmp/prof_x/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
PY
