#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): SQ issue/stall counters of one workload, one rocprofv3 pass per counter group.
# usage: scripts/pmc_sq.sh <tag> "<perf_quick args>"   -> gpurun_out/sq_<tag>.txt
set -u
TAG=${1:-min}
ARGS=${2:-"1e7 min 21 11 2"}
REPO=$(pwd)
mkdir -p "$REPO/gpurun_out"
OUT=$REPO/gpurun_out/sq_$TAG.txt
: > "$OUT"
cd /tmp && export TMPDIR=/tmp
GROUPS_=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY"
         "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_SALU"
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_THREAD_CYCLES_VALU"
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_IFETCH SQ_WAVE32_INSTS GRBM_GUI_ACTIVE")
i=0
for g in "${GROUPS_[@]}"; do
  rm -rf /tmp/sq_$i
  timeout 300 rocprofv3 --pmc $g --output-format csv -d /tmp/sq_$i -- python $REPO/scripts/perf_quick.py $ARGS > /tmp/sq_$i.log 2>&1
  f=$(find /tmp/sq_$i -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then
    python - "$f" >> "$OUT" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
disp = collections.Counter(r["Kernel_Name"][:60] for r in rows)
for k, d in acc.items():
    if "bsk" not in k: continue
    ncnt = len(d)
    nd = disp[k] / max(ncnt, 1)
    print(k, "dispatches", int(nd), {c: v / nd for c, v in d.items()})
PY
  else
    echo "group $i failed: $(tail -2 /tmp/sq_$i.log)" >> "$OUT"
  fi
  i=$((i+1))
done
cat "$OUT"
