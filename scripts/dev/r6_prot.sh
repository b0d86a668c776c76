#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
R=$PWD
for v in base wide base wide; do
  lib=$R/bio_amd/csrc/libbiosketch.so; [ $v = wide ] && lib=$R/scripts/variants/libbsk_protwide.so
  echo "== $v"; BSK_LIB=$lib python bench.py --workload protmin --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['roofline'].get('kernel'), d['ms_per_step'], d['roofline'].get('frac'), d.get('config',{}).get('digest'))"
done
for v in base wide; do
  lib=$R/bio_amd/csrc/libbiosketch.so; [ $v = wide ] && lib=$R/scripts/variants/libbsk_protwide.so
  rm -rf /tmp/pmc_$v
  (cd /tmp && BSK_LIB=$lib timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_$v -- python $R/bench.py --workload protmin --steps 3 --warmup 1 > /dev/null 2>&1)
  python - <<PY
import csv,glob
f=glob.glob('/tmp/pmc_$v/**/*counter_collection.csv',recursive=True)[0]
tot={}
n={}
for r in csv.DictReader(open(f)):
    if 'prot_minimizer_fast' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE':
        tot.setdefault(r['Dispatch_Id'],[]).append(float(r['Counter_Value']))
v=[sum(x)/len(x) for x in tot.values()]
print("$v", "FETCH_SIZE KiB per dispatch x 2 (gfx950) -> GB:", [round(x*2*1024/1e9,2) for x in v])
PY
done
