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


def par_read(path, threads=3, piece=0):
    try:
        rd = fastx.ParallelReader(path, threads, piece)
    except fastx.FastxError as e:
        return [], e, None
    seqs, err = [], None
    try:
        for seq, offs in rd.pieces():
            seqs += [seq[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(len(offs) - 1)]
    except fastx.FastxError as e:
        err = e
    info = rd.info()
    rd.close()
    return seqs, err, info


def same_error(a, b):
    return (a is None and b is None) or (a is not None and b is not None and (a is b or getattr(a, "code", 1) == getattr(b, "code", 2)))


def test_block_parallel_reader_gives_the_serial_readers_records(tmp_path, monkeypatch):
    """bsk_fastx_par_*: every edge case and the reference's fixtures, with pieces of 1 byte ... 1 MiB (a piece boundary at every
    byte of every file) and tiny read windows: the same sequences in the same order, and the same error, as the serial reader."""
    files = [os.path.join(GOLD, f) for f in ("test.fa", "test.fq", "test2.fq", "test3.fq", "test4.fa", "blank.fx", "blank1.fx", "empty.fx")]
    for name, data in CASES.items():
        p = tmp_path / name
        p.write_bytes(data)
        files.append(str(p))
    for path in files:
        recs, _, err = read_all(path)
        want = [s for _, s, _ in recs]
        for piece in (1, 2, 3, 5, 16, 37, 100, 1 << 20):
            for buf in ("1", "3", "64", ""):
                if buf:
                    monkeypatch.setenv("BSK_FASTX_BUF", buf)
                else:
                    monkeypatch.delenv("BSK_FASTX_BUF", raising=False)
                got, perr, info = par_read(path, 3, piece)
                assert got == want, (path, piece, buf, len(got), len(want))
                assert same_error(err, perr), (path, piece, buf, err, perr)


def test_block_parallel_reader_on_generated_fastq(tmp_path):
    """4-line FASTQ with qualities full of '@', '+', '>' (quality lines starting with '@' are where a guess can go wrong), FASTA with
    '>' inside lines, multi-line FASTQ (no guess passes: every piece is re-parsed serially) and CRLF: pieces of many sizes."""
    rng = random.Random(11)

    def fastq(nrec, width=0, crlf=False):
        out = bytearray()
        nl = b"\r\n" if crlf else b"\n"
        for i in range(nrec):
            n = rng.choice([0, 1, 30, 75, 150])
            s = bytes(rng.choice(b"ACGTN") for _ in range(n))
            q = bytes(rng.choice(b"@@+>I5#") for _ in range(n))
            if width:
                s = nl.join(s[j:j + width] for j in range(0, max(n, 1), width))
                q = nl.join(q[j:j + width] for j in range(0, max(n, 1), width))
            out += b"@r%d @x>y" % i + nl + s + nl + b"+" + nl + q + nl
        return bytes(out)

    fa = bytearray()
    for i in range(400):
        s = bytes(rng.choice(b"ACGT>@") for _ in range(rng.choice([0, 5, 61, 200])))
        fa += b">s%d a>b\n" % i + b"\n".join(s[j:j + 60] for j in range(0, max(len(s), 1), 60)) + b"\n"
    for name, data, expect_reparse in (("a.fq", fastq(2000), False), ("crlf.fq", fastq(500, crlf=True), False), ("ml.fq", fastq(300, width=40), True),
                                       ("a.fa", bytes(fa), False)):
        p = tmp_path / name
        p.write_bytes(data)
        recs, _, err = read_all(str(p))
        assert err is None
        want = [s for _, s, _ in recs]
        for piece in (64, 1000, 4096, 50000, 0):
            for threads in (1, 4):
                got, perr, info = par_read(str(p), threads, piece)
                assert perr is None and got == want, (name, piece, threads, len(got), len(want))
                if piece == 4096 and not expect_reparse:
                    assert info["reparsed_pieces"] <= 2, (name, info)  # the guess is right nearly always
    # a gzip file is not for this reader
    gz = tmp_path / "z.fq.gz"
    with gzip.open(gz, "wb") as g:
        g.write(fastq(10))
    with pytest.raises(fastx.FastxError) as ei:
        fastx.ParallelReader(str(gz), 2)
    assert ei.value.code == L.ERR_UNSUPPORTED


bgzf_compress = FO.bgzf_compress


def test_block_parallel_reader_reads_bgzf(tmp_path):
    """A BGZF file (bgzip: blocked gzip whose members record their own size) is a gzip file the block-parallel reader CAN take: its
    members are located without inflating and inflated independently, pieces are ranges of the uncompressed text.  Same records as the
    serial reader (which reads it as the multi-member gzip it is); an ordinary gzip file stays with the serial reader."""
    rng = random.Random(17)
    text = bytearray()
    for i in range(3000):
        n = rng.choice([0, 1, 30, 75, 150, 400])
        s = bytes(rng.choice(b"ACGTN") for _ in range(n))
        q = bytes(rng.choice(b"@@+>I5#") for _ in range(n))
        text += b"@r%d x@y\n" % i + s + b"\n+\n" + q + b"\n"
    text = bytes(text)
    plain = tmp_path / "r.fq"
    plain.write_bytes(text)
    want = [s for _, s, _ in read_all(str(plain))[0]]
    assert len(want) == 3000
    for name, blob in (("big.fq.gz", bgzf_compress(text)), ("ragged.fq.gz", bgzf_compress(text, 700, 1, rng)),
                       ("tiny.fq.gz", bgzf_compress(text[:5000], 7, 6, rng))):
        p = tmp_path / name
        p.write_bytes(blob)
        assert gzip.decompress(blob) == (text if name != "tiny.fq.gz" else text[:5000])  # it IS a gzip file
        ser = [s for _, s, _ in read_all(str(p))[0]]
        for piece in (0, 100, 1777, 50000):
            for threads in (1, 4):
                got, perr, info = par_read(str(p), threads, piece)
                assert perr is None and got == ser, (name, piece, threads, len(got), len(ser))
        if name != "tiny.fq.gz":
            assert ser == want
    # FASTA, and an empty BGZF file (only the end-of-file member)
    fa = b"".join(b">s%d\n%s\n" % (i, bytes(rng.choice(b"ACGT") for _ in range(rng.choice([0, 61, 200])))) for i in range(500))
    p = tmp_path / "a.fa.gz"
    p.write_bytes(bgzf_compress(fa, 333, 6, rng))
    assert par_read(str(p), 3, 500)[0] == [s for _, s, _ in read_all(str(p))[0]]
    p = tmp_path / "empty.fa.gz"
    p.write_bytes(bgzf_compress(b""))
    assert par_read(str(p), 2, 0)[:2] == ([], None)
    # a damaged block: an error, not silence
    blob = bytearray(bgzf_compress(text, 4000))
    blob[len(blob) // 2] ^= 0x55
    p = tmp_path / "bad.fq.gz"
    p.write_bytes(bytes(blob))
    got, perr, _ = par_read(str(p), 3, 20000)
    assert perr is not None and len(got) < 3000
