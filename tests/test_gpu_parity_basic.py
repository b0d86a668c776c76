"""GPU parity: HIP kernels (through the C ABI) vs the CPU oracle, bit-exact."""
import random

import numpy as np
import pytest

from bio_amd import _lib as L

pytestmark = pytest.mark.gpu


def rand_dna(rng, n, alpha="ACGT"):
    return "".join(rng.choice(alpha) for _ in range(n))


def check_minimizer(engine, oracle, seqs, k, w, circular=False):
    b = engine.batch(seqs)
    res = engine.run(b, engine.params(L.MINIMIZER, k, w=w, circular=circular))
    for i, s in enumerate(seqs):
        st, h, p = res.read(i)
        try:
            eh, ep, es, fl = oracle.minimizer(s, k, w, circular, closed=True)
        except oracle.OracleError as e:
            assert e.name == "ErrShortSeq"
            assert (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0, (i, s, k, w)
            continue
        assert (st & L.ST_CODE_MASK) == L.ST_OK, (i, s, st)
        assert np.array_equal(h, eh), (i, s, k, w, h, eh)
        assert np.array_equal(p & L.POS_MASK, ep), (i, s, k, w)
        assert np.array_equal(p >> 31, es), (i, s, k, w)
        assert (st & 0xF0) == fl, (i, s, k, w, st, fl)
    res.close()
    b.close()


def test_reference_known_answer(engine):
    # sketches/sketch_test.go:33-76 TestMinimizer
    from bio_amd import sketches as S
    seq, _ = S.NewSeq(S.DNA, "GGCAAGTTCGTCA")
    sk, err = S.NewMinimizerSketch(seq, 5, 3, False, engine)
    assert err is None
    codes, idxs = [], []
    while True:
        code, ok = sk.NextMinimizer()
        if not ok:
            break
        codes.append(code)
        idxs.append(sk.Index())
    assert codes == [973456138564179607, 2645801399420473919, 1099502864234245338, 6763474888237448943,
                     2737971715116251183]
    assert idxs == [0, 1, 4, 7, 8]


@pytest.mark.parametrize("k,w", [(21, 11), (5, 3), (31, 15), (15, 10), (1, 1), (7, 1), (3, 2), (33, 5), (64, 4), (70, 3),
                                 (21, 40), (21, 7), (21, 13), (21, 17), (25, 19), (21, 32), (15, 2), (15, 24), (9, 28), (21, 33)])
def test_minimizer_random(engine, oracle, k, w):
    rng = random.Random(k * 1000 + w)
    seqs = [rand_dna(rng, rng.choice([150, 150, 150, rng.randint(1, 300)])) for _ in range(300)]
    check_minimizer(engine, oracle, seqs, k, w)


def test_minimizer_adversarial(engine, oracle):
    rng = random.Random(5)
    seqs = ["A" * 150, "AC" * 75, "ACG" * 50, "ACGT" * 40, "T" * 31, "", "A", "ACGTTGCA" * 20,
            "GAATTC" * 30, rand_dna(rng, 150, "AC"), rand_dna(rng, 150, "AAAC")]
    for L_ in (20, 21, 30, 31, 32, 33):
        seqs.append(rand_dna(rng, L_))
    for k, w in [(21, 11), (5, 3), (11, 11), (4, 16)]:
        check_minimizer(engine, oracle, seqs, k, w)


def test_minimizer_non_acgt_and_lowercase(engine, oracle):
    rng = random.Random(7)
    seqs = [rand_dna(rng, 150, "ACGTN"), rand_dna(rng, 150, "acgtACGT"), rand_dna(rng, 200, "ACGTRYKMSWN"),
            rand_dna(rng, 150), "N" * 150, rand_dna(rng, 150, "ACGTU")]
    check_minimizer(engine, oracle, seqs, 21, 11)
    check_minimizer(engine, oracle, seqs, 5, 3)


def test_minimizer_circular(engine, oracle):
    rng = random.Random(9)
    seqs = [rand_dna(rng, n) for n in (150, 31, 40, 500, 29, 10)] + [rand_dna(rng, 100, "ACGTN")]
    check_minimizer(engine, oracle, seqs, 21, 11, circular=True)
    check_minimizer(engine, oracle, seqs, 7, 4, circular=True)


def test_minimizer_overflow_of_lds_staging(engine, oracle):
    # strictly "decreasing" runs force > 32 selections per read -> exercises the direct re-run path
    rng = random.Random(11)
    seqs = [rand_dna(rng, 400) for _ in range(70)]
    check_minimizer(engine, oracle, seqs, 9, 2)
    check_minimizer(engine, oracle, seqs, 15, 1)


def test_minimizer_paired_staging_columns(engine, oracle, monkeypatch):
    """k_minimizer_fast stages lanes l and l+32 in one LDS column filled from both ends: columns that overflow because of
    their SUM, lanes that run out of the column alone (both directions), and one-sided columns (BSK_NO_DENSE keeps these
    batches on that kernel)."""
    monkeypatch.setenv("BSK_NO_DENSE", "1")
    rng = random.Random(56)
    seqs = [rand_dna(rng, rng.randint(150, 330)) for _ in range(200)]  # 22..52 tuples per read: some column sums reach 56, some do not
    check_minimizer(engine, oracle, seqs, 21, 11)
    seqs = [rand_dna(rng, 150) for _ in range(128)]
    for lane in (3, 45, 64 + 31, 64 + 32):  # ~150 tuples: beyond the whole column, upwards (lane < 32) and downwards
        seqs[lane] = rand_dna(rng, 900)
    check_minimizer(engine, oracle, seqs, 21, 11)
    for lo, hi in ((10, 280), (280, 10), (0, 330), (330, 25)):  # one lane of every column (nearly) empty, the other up to ~48 tuples
        seqs = [rand_dna(rng, lo if (i & 32) == 0 else hi) for i in range(192)]
        check_minimizer(engine, oracle, seqs, 21, 11)
    seqs = [rand_dna(rng, rng.choice([150, 151, 170, 200])) for _ in range(700)]  # several tickets, ragged, w > 16 too
    for k, w in ((21, 11), (31, 15), (15, 20), (25, 32)):
        check_minimizer(engine, oracle, seqs, k, w)


@pytest.mark.parametrize("k,canonical,circular", [(21, True, False), (21, False, False), (10, True, True), (1, True, False),
                                                  (31, True, False), (65, True, False), (100, False, True)])
def test_nthash_stream(engine, oracle, k, canonical, circular):
    rng = random.Random(k)
    seqs = [rand_dna(rng, rng.choice([150, 150, rng.randint(1, 400)])) for _ in range(200)]
    seqs += ["", "A" * 200, rand_dna(rng, 150, "ACGTN")]
    b = engine.batch(seqs)
    res = engine.run(b, engine.params(L.NTHASH, k, canonical=canonical, circular=circular))
    for i, s in enumerate(seqs):
        st, h, _ = res.read(i)
        try:
            eh, _es = oracle.nthash(s, k, canonical, circular)
        except oracle.OracleError as e:
            assert e.name == "ErrShortSeq" and (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0
            continue
        assert np.array_equal(h, eh), (i, s, k)


def test_synth_batch_roundtrip_and_digest(engine, oracle):
    b = engine.synth(L.ALPHA_DNA, 5000, 150, 0x5EED0003)
    data, offs = b.fetch_ascii(0, 5000)
    assert len(data) == 5000 * 150 and set(np.unique(data).tolist()) <= set(b"ACGT")
    res = engine.run(b, engine.params(L.MINIMIZER, 21, w=11))
    nt, ck = oracle.batch_run(4, data, offs, 21, 11, threads=2)
    d = res.digest()
    assert d["n_tuples"] == nt and d["checksum"] == ck, (d, nt, ck)
    res2 = engine.run(b, engine.params(L.NTHASH, 21))
    nt2, ck2 = oracle.batch_run(2, data, offs, 21, 0, threads=2)
    d2 = res2.digest()
    assert d2["n_tuples"] == nt2 == 5000 * 130 and d2["checksum"] == ck2


def test_contexts_are_independent_across_threads(oracle):
    """SURVEY 8b threading: one context per worker, different contexts concurrently are safe (the reference's iterators are
    one-goroutine objects with pooled allocation as the only shared state)."""
    import threading
    from bio_amd import sketches as S
    rng = random.Random(99)
    sets = [[rand_dna(rng, rng.randint(30, 400)) for _ in range(300)] for _ in range(4)]
    errors = []

    def worker(seqs, k, w):
        try:
            eng = S.Engine(0)
            for _ in range(5):
                b = eng.batch(seqs)
                res = eng.run(b, eng.params(L.MINIMIZER, k, w=w))
                for i in (0, 7, 150, 299):
                    st, h, p = res.read(i)
                    try:
                        eh, ep, _, _ = oracle.minimizer(seqs[i], k, w, False, closed=True)
                    except oracle.OracleError:
                        assert (st & L.ST_CODE_MASK) == L.ST_SHORT
                        continue
                    assert np.array_equal(h, eh) and np.array_equal(p & L.POS_MASK, ep)
                res.close()
                b.close()
            eng.close() if hasattr(eng, "close") else None
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=worker, args=(sets[i], 15 + 2 * i, 5 + i)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert errors == []


def test_degenerate_batches(engine):
    """No reads, only empty reads, one base: every kind answers with empty results and SHORT statuses, sets and digests
    included; nothing launches with an empty grid."""
    for seqs in ([], [""], ["", "", ""], ["A"], ["A", "", "C"]):
        for alpha, kinds in ((L.ALPHA_DNA, [(L.MINIMIZER, dict(k=5, w=3)), (L.SYNCMER, dict(k=5, s=2)), (L.NTHASH, dict(k=3)),
                                            (L.KMER, dict(k=3)), (L.KMER, dict(k=3, canonical=False)), (L.SIMHASH, dict(k=8, m=4, scale=1)),
                                            (L.PROT_HASH, dict(k=2)), (L.PROT_MINIMIZER, dict(k=2, w=2)),
                                            (L.MINIMIZER, dict(k=5, w=3, circular=True))]),
                             (L.ALPHA_PROTEIN, [(L.PROT_HASH, dict(k=9)), (L.PROT_MINIMIZER, dict(k=9, w=5))])):
            b = engine.batch(seqs, alpha)
            for kind, pk in kinds:
                res = engine.run(b, engine.params(kind, **pk))
                assert res.info()["n_reads"] == len(seqs) and res.info()["n_tuples"] == 0
                d = res.digest()
                assert d["n_tuples"] == 0 and d["short"] == len(seqs)
                offs, vals = res.sets()
                assert len(offs) == len(seqs) + 1 and len(vals) == 0
                res.close()
            if alpha == L.ALPHA_DNA:
                t = b.translate(1, 1)
                assert t.info()["n_bases"] == 0
                t.close()
            b.close()


@pytest.mark.parametrize("k,w", [(21, 2), (21, 5), (15, 8), (31, 4)])
def test_dense_minimizers_and_their_fallback(engine, oracle, k, w):
    """Small windows select more than 32 positions per 150-bp read: the per-read-slab kernel (k_minimizer_dense) flushes
    mid-read; a low-complexity read (a new minimizer at every position) outgrows its slab and makes the call re-plan."""
    rng = random.Random(k * 7 + w)
    seqs = [rand_dna(rng, rng.choice([150, 150, 250, rng.randint(1, 300)])) for _ in range(300)]
    check_minimizer(engine, oracle, seqs, k, w)                                   # dense kernel
    check_minimizer(engine, oracle, seqs + ["A" * 250, "AC" * 100], k, w)        # slab overflow -> re-plan
    check_minimizer(engine, oracle, seqs + [rand_dna(rng, 200, "ACGTN")], k, w)  # mixed: dense main launch + ASCII side launch


def test_timed_entry_point_on_every_plan(engine):
    """bsk_sketch_timed (the bench's entry) must leave the same result as bsk_sketch whatever plan the batch takes:
    slab, dense, mixed, tiled, protein, translated."""
    rng = random.Random(31)
    reads = [rand_dna(rng, rng.choice([150, 250, rng.randint(30, 300)])) for _ in range(400)]
    with_n = reads[:399] + [rand_dna(rng, 200, "ACGTN")]
    contigs = [rand_dna(rng, n) for n in (6000, 300, 9000)]
    prot = ["".join(rng.choice("ACDEFGHIKLMNPQRSTVWY") for _ in range(rng.randint(40, 500))) for _ in range(200)]
    cases = [(reads, L.ALPHA_DNA, [(L.MINIMIZER, dict(k=21, w=11)), (L.MINIMIZER, dict(k=21, w=5)), (L.SYNCMER, dict(k=31, s=11)), (L.NTHASH, dict(k=21)),
                                   (L.SIMHASH, dict(k=21, m=5, scale=5)), (L.PROT_MINIMIZER, dict(k=9, w=5, frame=2))]),
             (with_n, L.ALPHA_DNA, [(L.MINIMIZER, dict(k=21, w=11)), (L.NTHASH, dict(k=21)), (L.SYNCMER, dict(k=21, s=11))]),
             (contigs, L.ALPHA_DNA, [(L.MINIMIZER, dict(k=21, w=11)), (L.NTHASH, dict(k=31)), (L.SYNCMER, dict(k=31, s=16))]),
             (prot, L.ALPHA_PROTEIN, [(L.PROT_MINIMIZER, dict(k=9, w=5)), (L.PROT_HASH, dict(k=10))])]
    for seqs, alpha, kinds in cases:
        b = engine.batch(seqs, alpha)
        for kind, pk in kinds:
            p = engine.params(kind, **pk)
            want = engine.run(b, p).digest()
            res, ms = engine.run_timed(b, p, 1, 2)
            assert len(ms) == 2 and all(m > 0 for m in ms)
            assert res.digest() == want, (kind, pk)
            res2, _ = engine.run_timed(b, p, 0, 1, reuse=res)   # the bench's pattern: repeat on the sized result
            assert res2.digest() == want, (kind, pk)


def test_minimizer_low_complexity_reads_take_the_exact_list(engine, oracle):
    """k_minimizer_pk sends READS with a 27-bit key tie to a list for the exact machine (per read, not per unit): batches where a
    third of the reads have poly-A / poly-G tails, dinucleotide repeats or are reverse-complement palindromes (real 64-bit ties),
    mixed with random reads in the same units, ragged lengths, several (k, w)."""
    rng = random.Random(77)
    seqs = []
    for i in range(700):
        L_ = rng.randint(60, 260)
        kind = i % 6
        if kind == 0:
            s = rand_dna(rng, L_ // 2) + "A" * (L_ - L_ // 2)
        elif kind == 1:
            s = "G" * (L_ // 3) + rand_dna(rng, L_ - L_ // 3)
        elif kind == 2:
            s = ("AC" * L_)[:L_]
        elif kind == 3:
            h = rand_dna(rng, L_ // 2)
            s = h + h[::-1].translate(str.maketrans("ACGT", "TGCA"))  # its own reverse complement
        else:
            s = rand_dna(rng, L_)
        seqs.append(s)
    for k, w in ((21, 11), (15, 10), (9, 5), (31, 13)):
        check_minimizer(engine, oracle, seqs, k, w)
