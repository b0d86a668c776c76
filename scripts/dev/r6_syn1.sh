#!/bin/bash
# round 6, first GPU call: the fused-emit syncmer kernel -- parity, then bench-size A/B against k_syncmer_pk; the ADVICE fixes' tests
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_syn1
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity_kinds.py -x -q -m gpu -k "syncmer" > $O/pytest_syn.txt 2>&1; tail -5 $O/pytest_syn.txt
for v in "" "BSK_NO_SYN_PF=1"; do
  echo "== $v" >> $O/perf.txt
  env $v timeout 300 python scripts/perf_quick.py 1.25e8 syn 31 11 6 >> $O/perf.txt 2>&1
  env $v timeout 300 python scripts/perf_quick.py 9e7 syn 31 11 6 200 >> $O/perf.txt 2>&1
done
cat $O/perf.txt | grep -v synth
timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_class_plans.py tests/test_gpu_comm.py -x -q -m gpu > $O/pytest_adv.txt 2>&1; tail -8 $O/pytest_adv.txt
