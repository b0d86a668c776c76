"""FASTA/Q reader of the library (bio_amd/csrc/fastx.cpp, host only) vs the reference's reader semantics.

CPU tests: the reader is host code; no GPU call is made here (bsk_batch_from_fastx is covered by tests/test_gpu_fastx.py).
"""
import gzip
import os
import random

import pytest

from bio_amd import _lib as L
from bio_amd import fastx
from oracle import fastx_oracle as FO

GOLD = os.path.join(os.path.dirname(__file__), "golden", "fastx")


def read_all(path, chunk=0):
    r = fastx.Reader(path)
    recs = []
    err = None
    try:
        for c in r.chunks(chunk):
            for i in range(len(c)):
                recs.append((c.name(i), c.sequence(i), c.quality(i)))
    except fastx.FastxError as e:
        err = e
    isq = r.IsFastq if recs else None
    r.Close()
    return recs, isq, err


def same_as_oracle(path):
    data = open(path, "rb").read()
    orecs, oq, oerr = FO.read_records(data)
    recs, isq, err = read_all(path)
    assert recs == orecs, (path, len(recs), len(orecs))
    if oerr is None:
        assert err is None
    elif isinstance(oerr, FO.NotFastx):
        assert err is fastx.ErrNotFASTXFormat
    else:
        assert err is fastx.ErrBadFASTQFormat
    return recs


def test_reference_reader_tests():
    """reader_test.go: TestFastaReader (6 records), TestFastqReader (8), TestFastqReader2 (5), TestFastqReader3 (3 records of equal
    length although quality lines start with '@'), TestBlankFile (ErrNotFASTXFormat), TestBlankFile2 / TestEmptyFile (nothing)."""
    assert len(same_as_oracle(os.path.join(GOLD, "test.fa"))) == 6
    assert len(same_as_oracle(os.path.join(GOLD, "test.fq"))) == 8
    assert len(same_as_oracle(os.path.join(GOLD, "test2.fq"))) == 5
    r3 = same_as_oracle(os.path.join(GOLD, "test3.fq"))
    assert len(r3) == 3 and len({len(s) for _, s, _ in r3}) == 1 and all(len(s) == len(q) for _, s, q in r3)
    recs, _, err = read_all(os.path.join(GOLD, "blank.fx"))
    assert recs == [] and err is fastx.ErrNotFASTXFormat
    for f in ("blank1.fx", "empty.fx"):
        recs, _, err = read_all(os.path.join(GOLD, f))
        assert recs == [] and err is None
    recs = same_as_oracle(os.path.join(GOLD, "test4.fa"))
    assert [(n, s) for n, s, _ in recs] == [(b"a", b"ATC"), (b"b", b""), (b"123", b"ATCGN"), (b"abcdefg", b"ATCGNGCCTN")]
    fa = same_as_oracle(os.path.join(GOLD, "test.fa"))
    assert fa[0][0] == b"test <trap> ab > cd"           # a '>' inside a header line is data
    assert fa[3] == (b"record with no sequence", b"", None)


CASES = {
    "crlf.fa": b">a desc\r\nACGT\r\nTTGA\r\n>b\r\nGG\r\n",
    "no_final_newline.fa": b">a\nACGT\n>b\nTT",
    "header_only_at_eof.fa": b">a\nACGT\n>b",
    "leading_newlines.fq": b"\n\n@r1\nACGT\n+\nIIII\n",
    "inline_delims.fa": b">a>b >c\nAC>GT\n>d\nA\n",
    "multiline.fq": b"@r1\nACGT\nTTAA\n+r1\nIIII\n@@@@\n@r2\nGG\n+\n@I\n",
    "at_quality.fq": b"@r1\nACGTACGT\n+\n@IIIIIII\n@r2\nAC\n+\n@@\n@r3\nT\n+\n@\n",
    "plus_in_seq_pos.fq": b"@r1\nACGT\n+\n+III\n@r2\nAAAA\n+\nIIII\n",
    "empty_first_line.fa": b">\nACGT\n>b\nTT\n",
    "empty_record_ends_file.fa": b">a\nAC\n>\n>c\nGG\n",
    "qual_longer.fq": b"@r1\nAC\n+\nIIII\n@r2\nAC\n+\nII\n",
    "qual_shorter_at_eof.fq": b"@r1\nACGT\n+\nII\n",
    "no_plus_line.fq": b"@r1\nACGT\n@r2\nAC\n+\nII\n",
    "not_fastx.txt": b"hello\n>a\nAC\n",
    "cr_first.fa": b"\r\n>a\nAC\n",
    "protein.fa": b">p1\nMKVLAAGIVGLLLAQW\n>p2\nMSTNPKPQRKTKRNTNRRPQDVKFPGG\n",
    "spaces_kept.fa": b">a\nAC GT\nTT-A\n",
    # found by scripts/fuzz_fastx.py: an error that follows good records of the same chunk must not be lost
    "error_after_records.fq": b"@r0\nCGGTTCAGGCGNAATNN\n+\n>#I@@55I+@+55@55>\n@r1\n\n+r1\n\n@@@\nNTCANTTCTNGTCNNCN\n+\n+#@##@#+@##III5@\n@@@\n\n+@@\n\n",
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_edge_cases_equal_the_reference_restatement(tmp_path, name):
    p = tmp_path / name
    p.write_bytes(CASES[name])
    same_as_oracle(str(p))


def test_specific_expectations(tmp_path):
    def recs(data):
        p = tmp_path / "x"
        p.write_bytes(data)
        return read_all(str(p))
    r, isq, err = recs(CASES["multiline.fq"])
    assert isq and err is None and r == [(b"r1", b"ACGTTTAA", b"IIII@@@@"), (b"r2", b"GG", b"@I")]
    r, _, err = recs(CASES["at_quality.fq"])
    assert err is None and [x[2] for x in r] == [b"@IIIIIII", b"@@", b"@"]
    r, _, err = recs(CASES["qual_longer.fq"])
    assert r == [] and err is fastx.ErrBadFASTQFormat
    r, _, err = recs(CASES["crlf.fa"])
    assert r == [(b"a desc", b"ACGTTTGA", None), (b"b", b"GG", None)]
    r, _, err = recs(CASES["spaces_kept.fa"])
    assert r == [(b"a", b"AC GTTT-A", None)]          # the reader does not strip blanks inside lines (validation is a separate step)
    rd = fastx.Reader(str(tmp_path / "x"))
    rec, e = rd.Read()
    assert e is None and rec.ID == b"a" and rec.Name == b"a" and rec.Seq.Seq == b"AC GTTT-A"
    assert rd.Read() == (None, fastx.EOF)


def test_alphabet_guess_and_record_api(tmp_path):
    p = tmp_path / "p.fa"
    p.write_bytes(CASES["protein.fa"])
    rd, err = fastx.NewDefaultReader(str(p))
    assert err is None
    rec, e = rd.Read()
    assert e is None and rec.ID == b"p1" and rd.alphabet == L.ALPHA_PROTEIN
    rd.Close()
    for data, want in ((b">a\nACGTNNRY\n", L.ALPHA_DNA), (b">a\nACGUUU\n", L.ALPHA_RNA), (b">a\nACGTN-.\n", L.ALPHA_DNA_PLAIN),
                       (b">a\nACGURY\n", L.ALPHA_RNA_REDUNDANT), (b">a\nACGTU\n", L.ALPHA_PROTEIN), (b">a\nACGT12\n", -1), (b">a\n>b\nAC\n", -1)):
        q = tmp_path / "g.fa"
        q.write_bytes(data)
        rd = fastx.Reader(str(q))
        rd.read_chunk(1)
        assert rd.alphabet == want, data
        rd.Close()
    rd, err = fastx.NewDefaultReader(str(tmp_path / "missing.fa"))
    assert rd is None and isinstance(err, OSError)


def test_window_boundaries_gzip_and_chunking(tmp_path, monkeypatch):
    """Tiny read windows (BSK_FASTX_BUF) put every boundary inside records; gzip input; chunk limits."""
    rng = random.Random(4)
    recs = []
    fq = bytearray()
    for i in range(300):
        n = rng.choice([0, 1, 2, 50, 151])
        s = "".join(rng.choice("ACGTN") for _ in range(n)).encode()
        q = bytes(rng.choice(b"@+>IJ#5") for _ in range(n))       # qualities full of delimiter characters
        name = f"read{i} {'x' * rng.randint(0, 9)}@>".encode()
        recs.append((name, s, q))
        width = rng.choice([7, 60, 1000])
        lines = [s[j:j + width] for j in range(0, max(len(s), 1), width)]
        qlines = [q[j:j + width] for j in range(0, max(len(q), 1), width)]
        fq += b"@" + name + b"\n" + b"\n".join(lines) + b"\n+\n" + b"\n".join(qlines) + b"\n"
    # a quality line that starts with '+' right after the '+' line is still quality; a first quality line starting with '@' too
    plain = tmp_path / "r.fq"
    plain.write_bytes(bytes(fq))
    gz = tmp_path / "r.fq.gz"
    with gzip.open(gz, "wb") as g:
        g.write(bytes(fq))
    want, _, oerr = FO.read_records(bytes(fq))
    assert oerr is None and len(want) == 300
    for buf in ("1", "2", "3", "7", "64", "1000", ""):
        if buf:
            monkeypatch.setenv("BSK_FASTX_BUF", buf)
        else:
            monkeypatch.delenv("BSK_FASTX_BUF", raising=False)
        for path in (plain, gz):
            got, isq, err = read_all(str(path), chunk=rng.choice([0, 1, 17]))
            assert err is None and isq and got == want, (buf, str(path))
    rd = fastx.Reader(str(plain))
    sizes = [len(c) for c in rd.chunks(max_records=64)]
    assert sizes == [64, 64, 64, 64, 44]
    rd = fastx.Reader(str(plain))
    c = rd.read_chunk(max_bytes=1)
    assert len(c) >= 1 and int(c.offsets[-1]) >= 1     # at least one record, stops once the byte budget is reached


def test_truncated_gzip_is_an_io_error(tmp_path):
    """A damaged / truncated .gz must not end as a clean (shorter) file: the reference hands the read error to the caller
    (seqio/fastx/reader.go:262-268); here the good records of the chunk come first, then BSK_ERR_IO."""
    rng = random.Random(5)
    text = "".join("@r%d\n%s\n+\n%s\n" % (i, "".join(rng.choice("ACGT") for _ in range(100)), "I" * 100) for i in range(20000))
    blob = gzip.compress(text.encode(), 6)
    p = tmp_path / "cut.fq.gz"
    p.write_bytes(blob[: len(blob) // 2])  # cut in the middle of the deflate stream
    recs, _, err = read_all(str(p))
    assert err is not None and err.code == L.ERR_IO, err
    assert 0 < len(recs) < 20000
    for name, s, q in recs[:-1]:  # everything handed out before the error is a whole record
        assert len(s) == 100 and len(q) == 100
    # the intact file still reads clean
    p2 = tmp_path / "ok.fq.gz"
    p2.write_bytes(blob)
    recs, _, err = read_all(str(p2))
    assert err is None and len(recs) == 20000
