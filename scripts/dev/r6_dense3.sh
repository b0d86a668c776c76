#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
R=$PWD
for tk in 0 2 4 16 18 26; do
echo "== tk $tk"
rm -rf /tmp/prof_$tk
(cd /tmp && BSK_PFT_TK=$tk TOTAL=2e9 NSEQ=400 ONLY=minimizer timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tk -- python $R/scripts/dev/perf_long2.py 2>&1 | grep wall)
python - <<PY
import csv,glob
f=glob.glob('/tmp/prof_$tk/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r['Name'][:60], r['Calls'], r['AverageNs'], r['Percentage'])
PY
done

