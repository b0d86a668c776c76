#!/bin/bash
# dev: time + HBM traffic of library variants (scripts/variants/libbsk_<tag>.so) on one perf_quick.py workload.
# usage: scripts/exp_variants.sh "<tags>" "<perf_quick args>" [ENV=VAL ...]   -> gpurun_out/exp_<tag>.txt
TAGS=$1; ARGS=$2; shift 2
for e in "$@"; do export "$e"; done
REPO=$(pwd); mkdir -p gpurun_out
for t in $TAGS; do
  export BSK_LIB=$REPO/scripts/variants/libbsk_$t.so
  [ "$t" = "base" ] && export BSK_LIB=$REPO/bio_amd/csrc/libbiosketch.so
  python scripts/perf_quick.py $ARGS > gpurun_out/exp_$t.txt 2>&1
  bash scripts/pmc_traffic.sh exp_$t "$ARGS" > /dev/null 2>&1
  echo "== $t"; grep -E "Gbases|kernel ms|checksum" gpurun_out/exp_$t.txt; cat gpurun_out/traffic_exp_$t.txt
done
