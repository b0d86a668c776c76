"""Copy the DATA files the reference's reader tests read (seqio/fastx/reader_test.go: test.fa, test.fq, test2.fq,
test3.fq, test4.fa, blank.fx, blank1.fx, empty.fx) into tests/golden/fastx/.  They are inputs; the expectations of
reader_test.go (record counts 6 / 8 / 5 / 3, equal lengths in test3.fq, ErrNotFASTXFormat for blank.fx, no record for
blank1.fx and empty.fx) are restated in tests/test_fastx.py.   Run in the build container: python tests/golden/make_fastx_golden.py
"""
import os
import shutil

SRC = "/root/reference/seqio/fastx"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fastx")
os.makedirs(DST, exist_ok=True)
for f in ("test.fa", "test.fq", "test2.fq", "test3.fq", "test4.fa", "blank.fx", "blank1.fx", "empty.fx"):
    shutil.copyfile(os.path.join(SRC, f), os.path.join(DST, f))
    os.chmod(os.path.join(DST, f), 0o644)
    print(f, os.path.getsize(os.path.join(DST, f)))
