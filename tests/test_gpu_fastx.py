"""FASTA/Q file -> device batch -> sketch, end to end through the C ABI (bsk_batch_from_fastx)."""
import gzip
import os
import random

import numpy as np
import pytest

from bio_amd import _lib as L
from bio_amd import fastx
from oracle import fastx_oracle as FO

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "fastx")


def test_fastq_file_to_minimizers(engine, oracle, tmp_path):
    rng = random.Random(12)
    lines = []
    for i in range(1000):
        n = rng.choice([150, 150, 151, rng.randint(1, 300)])
        s = "".join(rng.choice("ACGT" if i % 7 else "ACGTN") for _ in range(n))
        lines.append(f"@r{i} sample\n{s}\n+\n{'I' * n}\n")
    data = "".join(lines).encode()
    path = tmp_path / "reads.fq.gz"
    with gzip.open(path, "wb") as g:
        g.write(data)
    want, _, _ = FO.read_records(data)
    rd = fastx.Reader(str(path))
    seen = 0
    while True:
        b, n = engine.batch_from_fastx(rd, max_records=300)
        if n == 0:
            break
        res = engine.run(b, engine.params(L.MINIMIZER, 21, w=11))
        for i in range(n):
            q = want[seen + i][1].decode()
            st, h, p = res.read(i)
            try:
                eh, ep, es, fl = oracle.minimizer(q, 21, 11, False, closed=True)
            except oracle.OracleError:
                assert (st & L.ST_CODE_MASK) == L.ST_SHORT
                continue
            assert np.array_equal(h, eh) and np.array_equal(p & L.POS_MASK, ep), (seen + i)
        seen += n
        res.close()
        b.close()
    assert seen == 1000 and rd.IsFastq and rd.alphabet == L.ALPHA_DNA_PLAIN  # seq.DNA: the first record is pure ACGT


def test_reference_fastq_and_protein_fasta(engine, oracle, tmp_path):
    rd = fastx.Reader(os.path.join(GOLD, "test.fq"))
    b, n = engine.batch_from_fastx(rd)
    assert n == 8
    want, _, _ = FO.read_records(open(os.path.join(GOLD, "test.fq"), "rb").read())
    res = engine.run(b, engine.params(L.NTHASH, 21))
    for i in range(n):
        _, h, _ = res.read(i)
        assert np.array_equal(h, oracle.nthash(want[i][1].decode(), 21, True)[0])
    assert engine.batch_from_fastx(rd) == (None, 0)
    # test.fa is RNA (U): seq.RNA by the reference's guess -- a nucleotide batch; hashed from ASCII with the published table
    rd = fastx.Reader(os.path.join(GOLD, "test.fa"))
    b, n = engine.batch_from_fastx(rd)
    assert n == 6 and rd.alphabet == L.ALPHA_RNA and b.info()["n_non_acgt_reads"] >= 1
    want, _, _ = FO.read_records(open(os.path.join(GOLD, "test.fa"), "rb").read())
    res = engine.run(b, engine.params(L.KMER, 11, canonical=False))  # second strand: RevComInplace with the RNA pairs (A <-> U)
    for i in range(n):
        if len(want[i][1]) >= 11:
            assert np.array_equal(res.read(i)[1], oracle.kmer_codes(want[i][1].decode(), 11, False, False, L.ALPHA_RNA)), i
        else:
            assert (res.read(i)[0] & L.ST_CODE_MASK) == L.ST_SHORT
    assert not np.array_equal(res.read(0)[1], oracle.kmer_codes(want[0][1].decode(), 11, False, False, L.ALPHA_DNA))
    p = tmp_path / "p.fa"
    p.write_bytes(b">p1\n" + b"MKVLAAGIVGLLLAQWERTYIPASDFGHKLCVNM" * 3 + b"\n>p2\nMSTNPKPQRKTKRNTNRRPQDVKFPGGGQIVGGVYLLPRRGPRLGVRATRK\n")
    rd = fastx.Reader(str(p))
    b, n = engine.batch_from_fastx(rd)
    assert n == 2 and rd.alphabet == L.ALPHA_PROTEIN
    res = engine.run(b, engine.params(L.PROT_HASH, 9))
    want, _, _ = FO.read_records(p.read_bytes())
    for i in range(2):
        _, h, _ = res.read(i)
        assert np.array_equal(h, oracle.protein_hashes(want[i][1], 9))


def test_read_hands_out_the_guessed_alphabet(engine, oracle, tmp_path):
    """Reader.Read() (reader.go:233, :430-435): the record's Seq carries the alphabet GUESSED from the file's first sequence -- the
    two-strand k-mer mode pairs letters with it (iterator.go:719), so a DNAredundant / RNA / RNAredundant file must not come out as
    plain DNA ('R' pairs with 'Y' in DNAredundant and stays 'R' in DNA)."""
    from bio_amd import sketches as S
    cases = {"dnar.fa": (b">a\nACGTRYACGTNNACGTKMACGTACGTAC\n>b\nACGTACGTACGTRACGTACGTYACG\n", S.DNAredundant, L.ALPHA_DNA),
             "rna.fa": (b">a\nACGUACGUACGUACGUACGUACGU\n>b\nUUUACGACGACGUUUACGACG\n", S.RNA, L.ALPHA_RNA),
             "rnar.fa": (b">a\nACGURYACGUNNACGUKMACGUACGUAC\n>b\nACGUACGUACGURACGUACGUYACG\n", S.RNAredundant, L.ALPHA_RNA_REDUNDANT),
             "dna.fa": (b">a\nACGTACGTACGTACGTACGTACGT\n>b\nTTTACGACGACGTTTACGACG\n", S.DNA, L.ALPHA_DNA_PLAIN)}
    for name, (text, want_alpha, code) in cases.items():
        p = tmp_path / name
        p.write_bytes(text)
        rd = fastx.Reader(str(p))
        recs = []
        while True:
            rec, err = rd.Read()
            if rec is None:
                break
            recs.append(rec)
        assert len(recs) == 2
        for rec in recs:
            assert rec.Seq.Alphabet is want_alpha, (name, rec.Seq.Alphabet)
            it, err = S.NewKmerIterator(rec.Seq, 7, False, False, engine)
            assert err is None
            got = []
            while True:
                c, ok, e = it.NextKmer()
                if not ok:
                    break
                got.append(c)
            want = oracle.kmer_codes(bytes(rec.Seq.Seq).decode(), 7, False, False, code)
            assert got == [int(x) for x in want], name
