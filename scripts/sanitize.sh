#!/bin/bash
# Sanitizer builds of the HOST side of libbiosketch.so (make SAN=asan|ubsan|tsan, bio_amd/csrc/Makefile) and the test runs against them.
#   scripts/sanitize.sh build            all three variants (here: hipcc cross-compiles; ~2 min each on 8 cores)
#   scripts/sanitize.sh cpu              the CPU suite's library tests under ASan and UBSan (build container)
#   scripts/sanitize.sh gpu <out.txt>    on a GPU box: pipeline + class-plan + fastx GPU tests under ASan / UBSan / TSan
# Results: profiles/r06/sanitizers.txt.  Python is not instrumented: the clang runtime is preloaded, leak detection is off
# (the interpreter's own allocations), and TSan only sees the library's own threads and locks (pipeline.cpp, fastx.cpp).
set -u
cd "$(dirname "$0")/.."
CSRC=bio_amd/csrc
rt() { make -s -C $CSRC san-runtime SAN=$1; }
run() {  # variant, options, tests...
    local v=$1; shift
    local opts=$1; shift
    echo "== $v: pytest $*"
    env LD_PRELOAD=$(rt $v) $opts BSK_LIB=$PWD/$CSRC/san-$v/libbiosketch.so python -m pytest "$@" -x -q -p no:cacheprovider 2>&1 | tail -n 15
}
case "${1:-}" in
build)
    for v in asan ubsan tsan; do make -j8 -C $CSRC SAN=$v libbiosketch.so > /tmp/san_$v.log 2>&1 || { tail -20 /tmp/san_$v.log; exit 1; }; echo "built san-$v"; done
    make -C $CSRC san-hostasan/libbiosketch.so > /tmp/san_hostasan.log 2>&1 || { tail -20 /tmp/san_hostasan.log; exit 1; }; echo "built san-hostasan" ;;
cpu)
    run asan "ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1" tests/test_fastx.py tests/test_abi_and_host.py
    run ubsan "UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1" tests/test_fastx.py tests/test_abi_and_host.py ;;
gpu)
    out=${2:-gpurun_out/sanitizers_gpu.txt}
    {
        # ASan on a GPU box: the host-code variant with gcc's runtime (the clang runtime dies at the first HIP allocation here: Makefile)
        echo "== hostasan: pytest -m gpu tests/test_gpu_pipeline.py tests/test_gpu_fastx.py   (not test_gpu_comm: librocm_smi throws a C++ exception inside RCCL init, which gcc libasan as a PRELOADED runtime cannot forward -- CHECK real___cxa_throw)"
        env LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 BSK_LIB=$PWD/$CSRC/san-hostasan/libbiosketch.so python -m pytest -m gpu tests/test_gpu_pipeline.py tests/test_gpu_fastx.py -x -q -s -p no:cacheprovider 2>&1 | grep -v "^\[Gloo\]\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -n 15
        run ubsan "UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1:suppressions=$PWD/scripts/ubsan.supp" -s -m gpu tests/test_gpu_pipeline.py tests/test_gpu_class_plans.py tests/test_gpu_fastx.py tests/test_gpu_long_sequences.py
        run tsan "TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0:suppressions=$PWD/scripts/tsan.supp" -m gpu tests/test_gpu_pipeline.py tests/test_gpu_class_plans.py
    } > "$out" 2>&1
    tail -n 40 "$out" ;;
*) echo "usage: $0 build|cpu|gpu [out]"; exit 2 ;;
esac
