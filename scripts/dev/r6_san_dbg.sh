#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_san
mkdir -p $O
CS=bio_amd/csrc
env LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 BSK_LIB=$PWD/$CS/san-hostasan/libbiosketch.so python -m pytest -m gpu tests/test_gpu_pipeline.py tests/test_gpu_fastx.py tests/test_gpu_comm.py -x -q -s -p no:cacheprovider > $O/hostasan.txt 2>&1; echo "rc=$?" >> $O/hostasan.txt
grep -n "ERROR: AddressSanitizer" -A25 $O/hostasan.txt | head -70
RT=$(make -s -C $CS san-runtime SAN=tsan)
env LD_PRELOAD=$RT TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0:suppressions=$PWD/scripts/tsan.supp BSK_LIB=$PWD/$CS/san-tsan/libbiosketch.so python -m pytest -m gpu tests/test_gpu_pipeline.py tests/test_gpu_class_plans.py -x -q -s -p no:cacheprovider > $O/tsan.txt 2>&1; echo "rc=$?" >> $O/tsan.txt
grep -c "WARNING: ThreadSanitizer" $O/tsan.txt; grep -n "WARNING: ThreadSanitizer" -A14 $O/tsan.txt | head -60; tail -3 $O/tsan.txt
