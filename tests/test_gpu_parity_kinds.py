"""GPU parity for the remaining iterator kinds: syncmer, k-mer codes, SimHash, protein hash / minimizer."""
import random

import numpy as np
import pytest

from bio_amd import _lib as L

pytestmark = pytest.mark.gpu


def rand_seq(rng, n, alpha="ACGT"):
    return "".join(rng.choice(alpha) for _ in range(n))


def dna_set(rng, n=200):
    seqs = [rand_seq(rng, rng.choice([150, 150, rng.randint(1, 320)])) for _ in range(n)]
    seqs += ["", "A" * 150, "AC" * 75, "ACGTTGCAACGT" * 12, rand_seq(rng, 150, "ACGTN"), rand_seq(rng, 150, "acgtACGT"),
             rand_seq(rng, 200, "ACGTRYKMSWBDHVN"), rand_seq(rng, 150, "AC")]
    return seqs


@pytest.mark.parametrize("k,s,circular", [(31, 11, False), (5, 2, False), (31, 16, False), (7, 7, False), (15, 14, False),
                                          (21, 1, False), (11, 5, True), (64, 33, False)])
def test_syncmer(engine, oracle, k, s, circular):
    rng = random.Random(1000 * k + s)
    seqs = dna_set(rng)
    b = engine.batch(seqs)
    res = engine.run(b, engine.params(L.SYNCMER, k, s=s, circular=circular))
    for i, q in enumerate(seqs):
        st, h, p = res.read(i)
        try:
            eh, ep, es, fl = oracle.syncmer(q, k, s, circular, closed=True)
        except oracle.OracleError as e:
            assert e.name == "ErrShortSeq" and (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0, (i, q, e.name)
            continue
        assert (st & L.ST_CODE_MASK) == L.ST_OK, (i, q)
        assert np.array_equal(h, eh), (i, q, k, s)
        assert np.array_equal(p & L.POS_MASK, ep) and np.array_equal(p >> 31, es), (i, q)
        assert (st & 0xF0) == fl, (i, q, st, fl)


def test_syncmer_matches_state_machine_too(engine, oracle):
    rng = random.Random(77)
    seqs = [rand_seq(rng, 150) for _ in range(100)]
    b = engine.batch(seqs)
    res = engine.run(b, engine.params(L.SYNCMER, 31, s=11))
    for i, q in enumerate(seqs):
        _, h, p = res.read(i)
        eh, ep, _, _ = oracle.syncmer(q, 31, 11)  # line-by-line restatement of sketch.go:312-477
        assert np.array_equal(h, eh) and np.array_equal(p & L.POS_MASK, ep)


def test_syncmer_invalid_s(engine):
    from bio_amd import sketches as S
    seq, _ = S.NewSeq(S.DNA, "ACGT" * 20)
    assert S.NewSyncmerSketch(seq, 5, 6, False, engine) == (None, S.ErrInvalidS)
    assert S.NewSyncmerSketch(seq, 5, 0, False, engine) == (None, S.ErrInvalidS)
    assert S.NewSyncmerSketch(seq, 0, 1, False, engine) == (None, S.ErrInvalidK)
    assert S.NewMinimizerSketch(seq, 5, 0, False, engine) == (None, S.ErrInvalidW)
    assert S.NewMinimizerSketch(S.Seq(S.DNA, "ACGTACG"), 5, 4, False, engine) == (None, S.ErrShortSeq)


@pytest.mark.parametrize("k,canonical,circular", [(10, True, False), (10, False, False), (31, True, False), (32, False, False),
                                                  (1, True, False), (5, True, True), (5, False, True)])
def test_kmer_codes(engine, oracle, k, canonical, circular):
    rng = random.Random(k)
    seqs = [rand_seq(rng, rng.choice([100, 150, rng.randint(1, 200)])) for _ in range(150)]
    seqs += ["", "A" * 40, rand_seq(rng, 100, "ACGTN"), rand_seq(rng, 100, "acgtRYKM"), "ACGTACGTXACGTACGTACGT" * 3,
             "X" + rand_seq(rng, 60), rand_seq(rng, 60) + "-"]
    b = engine.batch(seqs)
    res = engine.run(b, engine.params(L.KMER, k, canonical=canonical, circular=circular))
    for i, q in enumerate(seqs):
        st, h, _ = res.read(i)
        try:
            e = oracle.kmer_codes(q, k, canonical, circular)
        except oracle.OracleError as err:
            if err.name == "ErrShortSeq":
                assert (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0, (i, q)
            else:
                assert err.name == "ErrIllegalBase" and (st & L.ST_CODE_MASK) == L.ST_ILLEGAL, (i, q, err.name, st)
                # the codes before the first k-mer that contains the illegal base are still delivered
                q2 = (q + q[: k - 1]) if circular else q
                bad = min(j for j, c in enumerate(q2) if c.upper() not in "ACGTURYSWKMBDHVN")
                good = max(0, bad - k + 1)
                assert len(h) == good, (i, q, len(h), good)
                if good:
                    pre = oracle.kmer_codes(q2[: good + k - 1], k, True, False) if canonical else \
                        oracle.kmer_codes(q2[: good + k - 1], k, False, False)[:good]
                    assert np.array_equal(h, pre), (i, q)
            continue
        assert (st & L.ST_CODE_MASK) == L.ST_OK, (i, q, st)
        assert np.array_equal(h, e), (i, q, k, h[:4], e[:4])


def test_kmer_k_too_large(engine):
    from bio_amd import sketches as S
    seq, _ = S.NewSeq(S.DNA, "ACGT" * 20)
    assert S.NewKmerIterator(seq, 33, True, False, engine) == (None, S.ErrKTooLarge)


@pytest.mark.parametrize("k,m,scale,canonical", [(21, 5, 5, True), (31, 5, 5, True), (21, 5, 1, True), (21, 21, 1, False),
                                                 (16, 4, 13, True)])
def test_simhash(engine, oracle, k, m, scale, canonical):
    rng = random.Random(k * 100 + m)
    seqs = [rand_seq(rng, rng.choice([60, 150, rng.randint(1, 200)])) for _ in range(80)]
    seqs += ["GAACAATGTTCTCTAAAATTG", "GcACAATGTTCTCTAAAATTG", rand_seq(rng, 100, "ACGTN"), "A" * 100]
    b = engine.batch(seqs)
    res = engine.run(b, engine.params(L.SIMHASH, k, m=m, scale=scale, canonical=canonical))
    for i, q in enumerate(seqs):
        st, h, _ = res.read(i)
        try:
            e = oracle.simhash(q, k, m, scale, canonical)
        except oracle.OracleError as err:
            assert err.name == "ErrShortSeq" and (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0
            continue
        assert np.array_equal(h, e), (i, q, k, m, scale)


def test_simhash_argument_errors(engine):
    from bio_amd import sketches as S
    seq, _ = S.NewSeq(S.DNA, "ACGT" * 20)
    assert S.NewSimHashIterator(seq, 8, 3, 1, True, False, engine) == (None, S.ErrInvalidM)
    assert S.NewSimHashIterator(seq, 8, 5, 5, True, False, engine) == (None, S.ErrInvalidScale)


AA = "ACDEFGHIKLMNPQRSTVWY"


@pytest.mark.parametrize("k,w", [(9, 5), (10, 3), (3, 1), (2, 7), (33, 4)])
def test_protein_minimizer_and_hash(engine, oracle, k, w):
    rng = random.Random(k * 10 + w)
    seqs = [rand_seq(rng, rng.choice([300, rng.randint(1, 400)]), AA) for _ in range(120)]
    seqs += ["", "A" * 100, rand_seq(rng, 200, "AC"), rand_seq(rng, 120, AA + "X*")]
    b = engine.batch(seqs, L.ALPHA_PROTEIN)
    res = engine.run(b, engine.params(L.PROT_MINIMIZER, k, w=w))
    res2 = engine.run(b, engine.params(L.PROT_HASH, k))
    for i, q in enumerate(seqs):
        st, h, p = res.read(i)
        try:
            eh, ep, fl = oracle.protein_minimizer(q, k, w, closed=True)
        except oracle.OracleError as err:
            assert err.name == "ErrShortSeq" and (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0
        else:
            assert np.array_equal(h, eh) and np.array_equal(p & L.POS_MASK, ep) and (st & 0xF0) == fl, (i, q)
        st2, h2, _ = res2.read(i)
        try:
            e2 = oracle.protein_hashes(q, k)
        except oracle.OracleError as err:
            assert err.name == "ErrShortSeq" and (st2 & L.ST_CODE_MASK) == L.ST_SHORT and len(h2) == 0
        else:
            assert np.array_equal(h2, e2), (i, q)


@pytest.mark.parametrize("k", [9, 10, 11, 12, 13, 14, 15, 16, 17])
def test_protein_hash_register_kernel_all_k_and_long_sequences(engine, oracle, k):
    """k = 9..16 run on k_prot_hash_fast<K> (512-position chunks staged in LDS); 17 on the general kernel."""
    rng = random.Random(900 + k)
    lens = [3 * k - 1, 3 * k, 3 * k + 1, 47, 48, 49, 511, 512, 513, 512 + k - 1, 512 + k, 1023, 1024, 1025, 1500, 2600]
    seqs = [rand_seq(rng, n, AA) for n in lens] + [rand_seq(rng, rng.randint(1, 700), AA) for _ in range(150)]
    b = engine.batch(seqs, L.ALPHA_PROTEIN)
    res = engine.run(b, engine.params(L.PROT_HASH, k))
    for i, q in enumerate(seqs):
        st, h, _ = res.read(i)
        try:
            e = oracle.protein_hashes(q, k)
        except oracle.OracleError as err:
            assert err.name == "ErrShortSeq" and (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0
        else:
            assert np.array_equal(h, e), (i, len(q))


def test_protein_from_dna_is_refused_loudly(engine):
    from bio_amd import sketches as S
    seq, _ = S.NewSeq(S.DNA, "ACGT" * 30)
    with pytest.raises(S.DeviceError):  # DNA -> protein translation is not implemented: no silent fallback
        S.NewProteinIterator(seq, 5, 1, 1, engine)
