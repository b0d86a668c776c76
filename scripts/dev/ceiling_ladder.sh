#!/bin/bash
# dev (GPU box): the knock-out ceiling ladder at BENCH size (10^8 x 150 bp, k=21 w=11) for k_minimizer_pk and k_minimizer_ring, one box, one
# session.  Per variant: kernel rate (perf_quick: HIP events), GRBM_GUI_ACTIVE of the dispatch (one rocprofv3 --pmc pass -> average clock),
# board watts + sclk from rocm-smi while the same launch loops.  Variants are built by scripts/build_variant.sh (timing-only knock-outs:
# their results are wrong by construction).  usage: scripts/dev/ceiling_ladder.sh <out.json> [n]
OUT=${1:-gpurun_out/ceiling_ladder.json}; N=${2:-1e8}
REPO=$(pwd); mkdir -p "$(dirname "$OUT")"
export TMPDIR=/tmp
: > /tmp/ladder.jsonl
one() {  # kernel tag lib ring?
  KERN=$1; TAG=$2; LIB=$3
  if [ "$KERN" = ring ]; then export BSK_RING=1; unset BSK_NO_RING; else unset BSK_RING; export BSK_NO_RING=1; fi
  export BSK_LIB=$LIB
  # pass 1: timing (10 launches) with rocm-smi sampled while it loops
  python $REPO/scripts/perf_quick.py $N min 21 11 ${ITERS:-900} > /tmp/l_$TAG.log 2>&1 &
  PID=$!
  : > /tmp/l_$TAG.smi
  sleep ${SMI_DELAY:-3}
  while kill -0 $PID 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr -s ' ' | tr '\n' ';' >> /tmp/l_$TAG.smi; echo >> /tmp/l_$TAG.smi
    sleep 0.05
  done
  wait $PID
  # pass 2: cycles
  rm -rf /tmp/l_pmc
  (cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/l_pmc -- python $REPO/scripts/perf_quick.py $N min 21 11 4 > /tmp/l_pmc.log 2>&1)
  python - "$KERN" "$TAG" /tmp/l_$TAG.log /tmp/l_$TAG.smi "$(find /tmp/l_pmc -name '*counter_collection.csv' | head -1)" <<'PY' >> /tmp/ladder.jsonl
import sys, re, csv, json, collections
kern, tag, log, smi, pmc = sys.argv[1:6]
txt = open(log).read()
m = re.search(r"Gbases/s best=([\d.]+) avg=([\d.]+)", txt)
ms = re.search(r"kernel ms: \[([^\]]*)\]", txt)
msl = [float(x) for x in ms.group(1).split(",")] if ms else []
W, F = [], []
for line in open(smi):
    w = re.search(r"Power \(W\): ([\d.]+)", line); f = re.search(r"sclk clock level: \w+: \((\d+)Mhz\)", line)
    if w and f and float(w.group(1)) > 900: W.append(float(w.group(1))); F.append(int(f.group(1)))
cyc = None; dur = None; name = None
if pmc:
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(pmc)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and "k_minimizer" in r["Kernel_Name"] and "dense" not in r["Kernel_Name"]:
            rows[r["Kernel_Name"][:40]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), float(r["Counter_Value"])))
    for k_, v in rows.items():
        v.sort(); dur, cyc = v[len(v) // 2]; name = k_
n = 1e8 * 150
out = {"kernel": kern, "variant": tag, "gbases_best": float(m.group(1)) if m else None, "gbases_avg": float(m.group(2)) if m else None,
       "kernel_ms_median": sorted(msl)[len(msl) // 2] if msl else None,
       "frac_of_8TBs_if_real": None, "board_W": round(sum(W) / len(W), 1) if W else None, "sclk_MHz": round(sum(F) / len(F)) if F else None,
       "smi_samples": len(W), "digest": txt.strip().splitlines()[-1][:200], "grbm_cycles_xcd_sum": cyc, "pmc_dispatch_ms": dur / 1e6 if dur else None,
       "avg_clock_GHz_pmc": round(cyc / 8 / dur, 3) if cyc else None, "pmc_kernel": name}
print(json.dumps(out))
PY
  tail -1 /tmp/ladder.jsonl
}
V=$REPO/scripts/variants
if [ -n "$LADDER" ]; then
  for item in $LADDER; do  # kernel:tag:lib  (lib "base" = the shipped library)
    IFS=: read K T Lb <<< "$item"
    if [ "$Lb" = base ]; then one $K $T $REPO/bio_amd/csrc/libbiosketch.so; else one $K $T $V/libbsk_$Lb.so; fi
  done
else
one pk   shipped   $REPO/bio_amd/csrc/libbiosketch.so
one pk   notie     $V/libbsk_pnotie.so
one pk   nostage   $V/libbsk_pnostage.so
one pk   nostore   $V/libbsk_pnostore.so
one pk   all_three $V/libbsk_pnone.so
one ring shipped   $REPO/bio_amd/csrc/libbiosketch.so
one ring notie     $V/libbsk_rnotie.so
one ring nostage   $V/libbsk_rnostage.so
one ring nostore   $V/libbsk_rnostore.so
one ring all_three $V/libbsk_rnone.so
one pk   shipped_again $REPO/bio_amd/csrc/libbiosketch.so
fi
python - "$OUT" <<'PY'
import json, sys
rows = [json.loads(l) for l in open("/tmp/ladder.jsonl")]
alg = 31.935e9  # algorithmic bytes per launch of 10^8 reads (319.35 B/read)
for r in rows:
    if r["kernel_ms_median"]: r["frac_of_8TBs_if_real"] = round(alg / (r["kernel_ms_median"] * 1e-3) / 8e12, 4)
json.dump({"what": "knock-out ceiling ladder, 1e8 x 150 bp, k=21 w=11, one box, one session; knock-outs are timing-only builds (wrong results)",
           "algorithmic_bytes_per_launch": alg, "rows": rows}, open(sys.argv[1], "w"), indent=1)
PY
cat "$OUT" | head -5
