#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_syn2
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity_kinds.py -x -q -m gpu -k "syncmer" > $O/pytest_syn.txt 2>&1; tail -3 $O/pytest_syn.txt
for a in "1.25e8 syn 31 11 6" "9e7 syn 31 11 6 200" "1.25e8 syn 21 9 6 130" "6e7 syn 64 46 6 224"; do
  timeout 300 python scripts/perf_quick.py $a 2>&1 | grep -v synth >> $O/perf.txt
done
cat $O/perf.txt
