#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_syn4
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity_kinds.py tests/test_gpu_state_machine.py tests/test_gpu_binning.py tests/test_gpu_long_sequences.py -x -q -m gpu -k "syncmer or syn" > $O/pytest_syn.txt 2>&1; tail -3 $O/pytest_syn.txt
for a in "1.25e8 syn 31 11 6" "9e7 syn 31 11 6 200" "6e7 syn 31 11 6 250" "5e7 syn 31 11 6 300" "4.3e7 syn 31 11 6 350" "4e7 syn 31 11 6 380" "1e8 syn 35 11 6 150"; do
  timeout 300 python scripts/perf_quick.py $a 2>&1 | grep -E "Gbases|plan|checksum" >> $O/perf.txt
done
echo "== BSK_NO_SYN_PF" >> $O/perf.txt
for a in "6e7 syn 31 11 6 250" "4.3e7 syn 31 11 6 350" "1e8 syn 35 11 6 150"; do
  BSK_NO_SYN_PF=1 timeout 300 python scripts/perf_quick.py $a 2>&1 | grep -E "Gbases|plan|checksum" >> $O/perf.txt
done
cat $O/perf.txt
