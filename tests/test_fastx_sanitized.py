"""The record reader (bio_amd/csrc/fastx.cpp: serial, block-parallel and BGZF readers -- threads, raw buffers, zlib) under
AddressSanitizer + UndefinedBehaviorSanitizer: `make -C bio_amd/csrc san-fastx/libbsk_fastx.so` (g++; the reader has no HIP call) and
tests/test_fastx.py -- the replay of the reference's seqio/fastx/reader_test.go fixtures, the gzip / BGZF / block-parallel cases and the
malformed-input cases -- run again against that library in a child process with the sanitizer runtime preloaded.  Any report aborts the
child (-fno-sanitize-recover, ASan's default halt_on_error).  SURVEY.md section 5 "sanitizers"; VERDICT round 5 item 7."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bio_amd", "csrc")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_reader_fixtures_under_asan_and_ubsan():
    subprocess.check_call(["make", "-C", CSRC, "san-fastx/libbsk_fastx.so"], stdout=subprocess.DEVNULL)
    rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    assert os.path.isabs(rt) and os.path.exists(rt), rt
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               BSK_LIB=os.path.join(CSRC, "san-fastx", "libbsk_fastx.so"), BSK_LIB_PARTIAL="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_fastx.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    tail = r.stdout[-3000:] + r.stderr[-3000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
