# dev (GPU): every bench workload REPS times in fresh processes (the timed path may not re-size: a workload whose sizing is on a
# margin fails one run in a few -- minimizer250 did, NOTEBOOK 5.13), one line per run
for w in ${WORKLOADS:-minimizer250 minimizer400 syncmer250 syncmer minimizer protmin prothash nthash kmer simhash}; do
  for i in $(seq ${REPS:-4}); do
    out=$(timeout 300 python bench.py --workload $w --steps ${STEPS:-5} --warmup 1 --no-cpu-baseline --no-end-to-end --no-power-probe 2>&1 | tail -1)
    echo "$w run $i: $(echo "$out" | python -c 'import sys,json
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print("ok", d["value"], d["unit"], d["roofline"].get("kernel"))
except Exception: print("FAILED:", l[-300:])')"
  done
done
