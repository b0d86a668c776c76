#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_syn3
mkdir -p $O
for t in base pf_noexpand pf_noemit pf_nostore; do
  export BSK_LIB=$PWD/scripts/variants/libbsk_$t.so
  [ "$t" = "base" ] && export BSK_LIB=$PWD/bio_amd/csrc/libbiosketch.so
  echo "== $t" >> $O/perf.txt
  timeout 300 python scripts/perf_quick.py 1.25e8 syn 31 11 6 2>&1 | grep -E "Gbases|kernel ms" >> $O/perf.txt
done
cat $O/perf.txt
