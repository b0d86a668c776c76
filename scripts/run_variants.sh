#!/bin/bash
# dev: time library variants on one perf_quick workload.  usage: scripts/run_variants.sh "<tags>" "<perf_quick args>"
TAGS=$1; ARGS=$2
REPO=$(pwd); mkdir -p gpurun_out
for t in $TAGS; do
  export BSK_LIB=$REPO/scripts/variants/libbsk_$t.so
  [ "$t" = "base" ] && export BSK_LIB=$REPO/bio_amd/csrc/libbiosketch.so
  case $t in OLD*) export BSK_NO_PK=1;; *) unset BSK_NO_PK;; esac
  echo "== $t $(python scripts/perf_quick.py $ARGS 2>&1 | grep -E 'Gbases|plan:' | tr '\n' ' ')"
done
