"""Long sequences: tiled runs must reproduce the un-tiled iterator exactly (kernels_tile.hpp).

BSK_TILE_MIN / BSK_TILE_POS (read by the library at every call) shrink the tile threshold and the tile size so that
ordinary test sequences cross many tile seams, for every kind that tiles; then genuinely long sequences, including
one above 2^24 bases, run with the defaults.
"""
import os
import random

import numpy as np
import pytest

from bio_amd import _lib as L

pytestmark = pytest.mark.gpu


def rand_seq(rng, n, alpha="ACGT"):
    return "".join(rng.choice(alpha) for _ in range(n))


@pytest.fixture
def tiny_tiles():
    saved = {k: os.environ.get(k) for k in ("BSK_TILE_MIN", "BSK_TILE_POS")}

    def set_(tile_min, tile_pos):
        os.environ["BSK_TILE_MIN"] = str(tile_min)
        if tile_pos:
            os.environ["BSK_TILE_POS"] = str(tile_pos)
        else:
            os.environ.pop("BSK_TILE_POS", None)

    yield set_
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def seq_set(rng, alpha, k_hint):
    lens = [1, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 100, 127, 128, 129, 150, 255, 256, 257, 300, 511, 512, 513, 777, 1023, 1024, 1025,
            2000, 3333, k_hint - 1, k_hint, k_hint + 1, 2 * k_hint, 2 * k_hint + 1]
    seqs = [rand_seq(rng, n, alpha) for n in lens if n > 0] + [rand_seq(rng, rng.randint(1, 2500), alpha) for _ in range(60)]
    seqs += ["", "A" * 700, "AC" * 400, "ACGTTGCAACGT" * 70]  # ties everywhere: leftmost-wins must survive the seams
    return seqs


def check_sketch(engine, oracle, seqs, kind, fn, **pk):
    b = engine.batch(seqs)
    res = engine.run(b, engine.params(kind, **pk))
    for i, q in enumerate(seqs):
        st, h, p = res.read(i)
        try:
            eh, ep, es, fl = fn(q)
        except oracle.OracleError as e:
            assert e.name == "ErrShortSeq" and (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0, (i, len(q), e.name, st)
            continue
        assert (st & L.ST_CODE_MASK) == L.ST_OK, (i, len(q), st)
        assert len(h) == len(eh), (i, len(q), len(h), len(eh))
        assert np.array_equal(h, eh), (i, len(q))
        assert np.array_equal(p & L.POS_MASK, ep) and np.array_equal(p >> 31, es), (i, len(q))
        assert (st & 0xF0) == fl, (i, len(q), st, fl)
    res.close()
    b.close()


@pytest.mark.parametrize("tile_pos", [16, 48, 0])
@pytest.mark.parametrize("k,w", [(21, 11), (31, 15), (5, 3), (7, 1), (21, 7), (64, 33), (15, 4)])
def test_minimizer_tiled(engine, oracle, tiny_tiles, k, w, tile_pos):
    tiny_tiles(40, tile_pos)
    rng = random.Random(k * 100 + w + tile_pos)
    for alpha in ("ACGT", "ACGTN"):
        seqs = seq_set(rng, alpha, k + w - 1)
        check_sketch(engine, oracle, seqs, L.MINIMIZER, lambda q: oracle.minimizer(q, k, w, False, closed=True), k=k, w=w)


@pytest.mark.parametrize("tile_pos", [16, 64, 0])
@pytest.mark.parametrize("k,s", [(31, 11), (31, 16), (5, 2), (15, 14), (21, 1), (64, 33), (11, 6), (9, 9), (31, 31)])
def test_syncmer_tiled(engine, oracle, tiny_tiles, k, s, tile_pos):
    tiny_tiles(40, tile_pos)
    rng = random.Random(k * 100 + s + tile_pos)
    for alpha in ("ACGT", "ACGTRYN"):
        seqs = seq_set(rng, alpha, 2 * k - s - 1)
        check_sketch(engine, oracle, seqs, L.SYNCMER, lambda q: oracle.syncmer(q, k, s, False, closed=True), k=k, s=s)


@pytest.mark.parametrize("tile_pos", [16, 0])
def test_streams_tiled(engine, oracle, tiny_tiles, tile_pos):
    tiny_tiles(40, tile_pos)
    rng = random.Random(5 + tile_pos)
    for alpha in ("ACGT", "ACGTNacgt"):
        for k, canonical in ((21, True), (21, False), (5, True), (5, False), (32, True), (32, False), (300, True)):
            seqs = seq_set(rng, alpha, k)
            b = engine.batch(seqs)
            rn = engine.run(b, engine.params(L.NTHASH, k, canonical=canonical))
            for i, q in enumerate(seqs):
                st, h, _ = rn.read(i)
                try:
                    e = oracle.nthash(q, k, canonical)[0]
                except oracle.OracleError:
                    assert (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0
                    continue
                assert np.array_equal(h, e), ("nthash", k, canonical, i, len(q))
            rn.close()
            if k <= 32:  # both NextKmer modes; canonical = False: two strands, the second from the reverse-complemented letters
                rk = engine.run(b, engine.params(L.KMER, k, canonical=canonical))
                for i, q in enumerate(seqs):
                    st, h, _ = rk.read(i)
                    try:
                        e = oracle.kmer_codes(q, k, canonical, False)
                    except oracle.OracleError as err:
                        assert err.name == "ErrShortSeq" and (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0
                        continue
                    assert np.array_equal(h, e), ("kmer", k, canonical, i, len(q))
                rk.close()
            b.close()
    seqs = seq_set(rng, "ACGT", 21)[:40]
    b = engine.batch(seqs)
    rs = engine.run(b, engine.params(L.SIMHASH, 21, m=5, scale=5))
    for i, q in enumerate(seqs):
        st, h, _ = rs.read(i)
        try:
            e = oracle.simhash(q, 21, 5, 5, True)
        except oracle.OracleError:
            assert (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0
            continue
        assert np.array_equal(h, e), ("simhash", i, len(q))


def test_kmer_illegal_base_stops_the_tiled_iterator_where_the_reference_stops(engine, oracle, tiny_tiles):
    tiny_tiles(40, 16)
    rng = random.Random(3)
    k = 11
    for badpos in (0, 5, 15, 16, 17, 40, 200, 777, 1499):
        q = list(rand_seq(rng, 1500))
        q[badpos] = "X"
        q = "".join(q)
        b = engine.batch([q, rand_seq(rng, 900)])
        res = engine.run(b, engine.params(L.KMER, k, canonical=True))
        st, h, _ = res.read(0)
        good = max(0, badpos - k + 1)
        assert (st & L.ST_CODE_MASK) == L.ST_ILLEGAL and len(h) == good, (badpos, st, len(h), good)
        if good:
            assert np.array_equal(h, oracle.kmer_codes(q[:badpos], k, True, False)[:good])
        st1, h1, _ = res.read(1)
        assert (st1 & L.ST_CODE_MASK) == L.ST_OK and len(h1) == 900 - k + 1


def test_circular_long_and_refusals(engine, oracle, tiny_tiles):
    tiny_tiles(40, 32)
    rng = random.Random(8)
    seqs = [rand_seq(rng, n) for n in (50, 333, 1000, 2049)]
    check_sketch(engine, oracle, seqs, L.MINIMIZER, lambda q: oracle.minimizer(q, 11, 5, True, closed=True), k=11, w=5, circular=True)
    b = engine.batch(seqs)
    rn = engine.run(b, engine.params(L.NTHASH, 9, circular=True))
    for i, q in enumerate(seqs):
        _, h, _ = rn.read(i)
        assert np.array_equal(h, oracle.nthash(q, 9, True, True)[0])


def test_circular_every_kmer_syncmer_keeps_its_own_length_rule_over_tiles(engine, oracle, tiny_tiles):
    """s == k syncmers run over tiles as the w = 1 minimizer, but refuse by the SYNCMER constructor's rule: circular, a sequence of
    k-1 bases passes len < 2k-s-1 (sketch.go:149) and its extended copy (2k-2 bases) can be hashed -- the minimizer would refuse it
    (found by the fuzz campaign, seed 101422)."""
    tiny_tiles(40, 16)
    rng = random.Random(21)
    k = 31
    seqs = [rand_seq(rng, n) for n in (k - 2, k - 1, k, k + 1, 300, 2000)]
    for circular in (True, False):
        check_sketch(engine, oracle, seqs, L.SYNCMER, lambda q: oracle.syncmer(q, k, k, circular, closed=True), k=k, s=k, circular=circular)
        check_sketch(engine, oracle, seqs, L.MINIMIZER, lambda q: oracle.minimizer(q, k, 1, circular, closed=True), k=k, w=1, circular=circular)


def test_long_sequences_with_default_tiles(engine, oracle):
    """1 Mbp (BASELINE configs[0]) and mixed contig lengths with the default tile size."""
    rng = np.random.default_rng(11)
    lens = [1_000_000, 5000, 4097, 4096, 150, 123_457, 20]
    seqs = ["".join(np.array(list("ACGT"))[rng.integers(0, 4, n)]) for n in lens]
    seqs[5] = seqs[5][:60_000] + "N" * 100 + seqs[5][60_100:]  # a gap: the whole batch runs on the ASCII kernels
    for batch_seqs in (seqs[:5], seqs):
        b = engine.batch(batch_seqs)
        rn = engine.run(b, engine.params(L.NTHASH, 21))
        rm = engine.run(b, engine.params(L.MINIMIZER, 21, w=11))
        rs = engine.run(b, engine.params(L.SYNCMER, 31, s=11))
        for i, q in enumerate(batch_seqs):
            st, h, _ = rn.read(i)
            if len(q) < 21:
                assert (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0
            else:
                assert np.array_equal(h, oracle.nthash(q, 21, True)[0]), ("nthash", i)
            st, h, p = rm.read(i)
            try:
                eh, ep, es, fl = oracle.minimizer(q, 21, 11, False, closed=True)
            except oracle.OracleError:
                assert (st & L.ST_CODE_MASK) == L.ST_SHORT
            else:
                assert np.array_equal(h, eh) and np.array_equal(p & L.POS_MASK, ep) and np.array_equal(p >> 31, es), ("min", i)
            st, h, p = rs.read(i)
            try:
                eh, ep, es, fl = oracle.syncmer(q, 31, 11, False, closed=True)
            except oracle.OracleError:
                assert (st & L.ST_CODE_MASK) == L.ST_SHORT
            else:
                assert np.array_equal(h, eh) and np.array_equal(p & L.POS_MASK, ep), ("syn", i)
        assert rm.digest()["n_tuples"] == rm.info()["n_tuples"]
        b.close()


def test_sequence_longer_than_2_pow_24(engine, oracle):
    n = (1 << 24) + 12345
    rng = np.random.default_rng(2)
    big = np.array(list(b"ACGT"), np.uint8)[rng.integers(0, 4, n)]
    small = np.frombuffer(b"ACGTTGCATGCATGCAAACCGGTTACGT", np.uint8)
    data = np.concatenate([small, big, small])
    offs = np.array([0, len(small), len(small) + n, len(small) * 2 + n], np.uint64)
    b = engine.batch_from_arrays(data, offs)
    q = big.tobytes().decode()
    rm = engine.run(b, engine.params(L.MINIMIZER, 21, w=11))
    st, h, p = rm.read(1)
    eh, ep, es, fl = oracle.minimizer(q, 21, 11, False, closed=True)
    assert len(h) == len(eh) and np.array_equal(h, eh) and np.array_equal(p & L.POS_MASK, ep) and np.array_equal(p >> 31, es)
    assert rm.info()["n_tuples"] == len(eh)  # the 28-base neighbours are too short for k = 21, w = 11
    rn = engine.run(b, engine.params(L.NTHASH, 31))
    st, h, _ = rn.read(1)
    assert len(h) == n - 30 and np.array_equal(h, oracle.nthash(q, 31, True)[0])
    rk = engine.run(b, engine.params(L.KMER, 21, canonical=False))  # two strands: 2(L-k+1) values, the second strand backwards
    st, h, _ = rk.read(1)
    assert len(h) == 2 * (n - 20) and np.array_equal(h, oracle.kmer_codes(q, 21, False, False))
    assert len(rk.read(0)[1]) == 2 * (28 - 20) and rk.info()["n_tuples"] == 2 * (n - 20) + 4 * 8
    rk.close()
    rs = engine.run(b, engine.params(L.SYNCMER, 15, s=15))  # s == k: every k-mer with its index (runs as the w = 1 minimizer over tiles)
    st, h, p = rs.read(1)
    eh, ep, es, fl = oracle.syncmer(q, 15, 15, False, closed=True)
    assert len(h) == n - 14 and np.array_equal(h, eh) and np.array_equal(p & L.POS_MASK, ep) and np.array_equal(p >> 31, es)
    # circular = true tiles too: the sequence with its first k-1 bases appended (iterator.go:642-646) is one more long sequence
    rc = engine.run(b, engine.params(L.NTHASH, 21, circular=True))
    st, h, _ = rc.read(1)
    assert len(h) == n and np.array_equal(h, oracle.nthash(q, 21, True, True)[0])
    rmc = engine.run(b, engine.params(L.MINIMIZER, 21, w=11, circular=True))
    st, h, p = rmc.read(1)
    eh, ep, es, fl = oracle.minimizer(q, 21, 11, True, closed=True)
    assert np.array_equal(h, eh) and np.array_equal(p & L.POS_MASK, ep) and np.array_equal(p >> 31, es)
    st, h, p = rmc.read(0)  # the 28-base neighbour stays ErrShortSeq: the length check is on the un-extended sequence (sketch.go:92)
    assert (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0


@pytest.mark.parametrize("tile_pos", [16, 48, 0])
def test_tiles_with_sparse_non_acgt_use_the_mixed_plan(engine, oracle, tiny_tiles, tile_pos):
    """Long sequences with a few N / IUPAC letters: only the tiles that touch them run on the ASCII kernels (per-word bits
    from the packer -> per-tile flags -> side launch); stream kinds write those tiles' runs in place so the sequence's run
    stays contiguous.  BSK_NO_MIXED gives the all-ASCII digest to compare with."""
    tiny_tiles(40, tile_pos)
    rng = random.Random(31 + tile_pos)
    seqs = []
    for n in (3000, 2500, 777, 5000, 64, 30):
        q = list(rand_seq(rng, n))
        for _ in range(max(1, n // 900)):
            q[rng.randrange(n)] = rng.choice("NRYn")
        seqs.append("".join(q))
    seqs.append(rand_seq(rng, 4000))  # one clean sequence
    check_sketch(engine, oracle, seqs, L.MINIMIZER, lambda q: oracle.minimizer(q, 21, 11, False, closed=True), k=21, w=11)
    check_sketch(engine, oracle, seqs, L.SYNCMER, lambda q: oracle.syncmer(q, 31, 11, False, closed=True), k=31, s=11)
    b = engine.batch(seqs)
    for kind, pk, fn in ((L.NTHASH, dict(k=21), lambda q: oracle.nthash(q, 21, True)[0]),
                         (L.KMER, dict(k=21), lambda q: oracle.kmer_codes(q, 21, True, False)),
                         (L.SIMHASH, dict(k=21, m=5, scale=5), lambda q: oracle.simhash(q, 21, 5, 5, True))):
        res = engine.run(b, engine.params(kind, **pk))
        for i, q in enumerate(seqs):
            _, h, _ = res.read(i)
            assert np.array_equal(h, fn(q)), (kind, i, len(q))
        d1 = res.digest()
        os.environ["BSK_NO_MIXED"] = "1"
        try:
            d2 = engine.run(b, engine.params(kind, **pk)).digest()
        finally:
            del os.environ["BSK_NO_MIXED"]
        assert d1 == d2


AA = "ACDEFGHIKLMNPQRSTVWY"


@pytest.mark.parametrize("tile_pos", [16, 48, 0])
def test_protein_sequences_tiled(engine, oracle, tiny_tiles, tile_pos):
    """Long protein sequences (and long translations) run as tiles too: the window closed form is the minimizer's, and
    the constructors' input-length rule is applied to the sequence, not to its tiles."""
    tiny_tiles(60, tile_pos)
    rng = random.Random(70 + tile_pos)
    seqs = [rand_seq(rng, n, AA) for n in (1, 26, 27, 31, 32, 61, 100, 300, 777, 2500)] + [rand_seq(rng, rng.randint(1, 900), AA) for _ in range(40)]
    seqs += ["A" * 500, "AC" * 300]
    b = engine.batch(seqs, L.ALPHA_PROTEIN)
    for k, w in ((9, 5), (10, 3), (3, 1), (12, 8), (33, 4)):
        rm = engine.run(b, engine.params(L.PROT_MINIMIZER, k, w=w))
        rh = engine.run(b, engine.params(L.PROT_HASH, k))
        for i, q in enumerate(seqs):
            st, h, p = rm.read(i)
            try:
                eh, ep, fl = oracle.protein_minimizer(q, k, w, closed=True)
            except oracle.OracleError:
                assert (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0, (k, w, i, len(q))
            else:
                assert np.array_equal(h, eh) and np.array_equal(p & L.POS_MASK, ep) and (st & 0xF0) == fl, (k, w, i, len(q))
            st, h, _ = rh.read(i)
            try:
                e = oracle.protein_hashes(q, k)
            except oracle.OracleError:
                assert (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0
            else:
                assert np.array_equal(h, e), (k, i, len(q))
    # DNA fed: the translation of a long contig is tiled, the length rule still looks at the nucleotides
    dna = [rand_seq(rng, n) for n in (26, 27, 29, 31, 100, 3000, 7001)]
    bd = engine.batch(dna)
    for frame in (1, -2):
        rm = engine.run(bd, engine.params(L.PROT_MINIMIZER, 9, w=5, codon_table=11, frame=frame))
        rh = engine.run(bd, engine.params(L.PROT_HASH, 9, codon_table=11, frame=frame))
        for i, q in enumerate(dna):
            st, h, p = rm.read(i)
            try:
                eh, ep, _ = oracle.protein_minimizer_nt(q, 9, 5, 11, frame)
            except oracle.OracleError:
                assert (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0, (frame, i)
            else:
                assert (st & L.ST_CODE_MASK) == L.ST_OK and np.array_equal(h, eh) and np.array_equal(p & L.POS_MASK, ep), (frame, i)
            st, h, _ = rh.read(i)
            try:
                e = oracle.protein_hashes_nt(q, 9, 11, frame)
            except oracle.OracleError:
                assert (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0
            else:
                assert np.array_equal(h, e), (frame, i)


def test_long_protein_default_tiles(engine, oracle):
    rng = random.Random(9)
    seqs = [rand_seq(rng, n, AA) for n in (35_000, 300, 4097, 12_000)]
    b = engine.batch(seqs, L.ALPHA_PROTEIN)
    rm = engine.run(b, engine.params(L.PROT_MINIMIZER, 9, w=5))
    rh = engine.run(b, engine.params(L.PROT_HASH, 10))
    for i, q in enumerate(seqs):
        _, h, p = rm.read(i)
        eh, ep, _ = oracle.protein_minimizer(q, 9, 5, closed=True)
        assert np.array_equal(h, eh) and np.array_equal(p & L.POS_MASK, ep), i
        assert np.array_equal(rh.read(i)[1], oracle.protein_hashes(q, 10)), i


def test_translation_of_a_chromosome_sized_sequence(engine, oracle):
    """Protein kinds fed DNA translate first (iterator-protein.go:62-67); a sequence of 2^24 bases or more is located by first word +
    length instead of the packed descriptor, and its translation (here more than 2^24 residues) runs as tiles."""
    n = 3 * (1 << 24) + 1000
    rng = np.random.default_rng(9)
    big = np.array(list(b"ACGT"), np.uint8)[rng.integers(0, 4, n)]
    small = np.frombuffer(b"ACGTTGCATGCATGCAAACCGGTTACGTAAGGCC", np.uint8)
    data = np.concatenate([small, big])
    b = engine.batch_from_arrays(data, np.array([0, len(small), len(small) + n], np.uint64))
    q = big.tobytes().decode()
    for frame in (1, -2):
        rm = engine.run(b, engine.params(L.PROT_MINIMIZER, 9, w=5, codon_table=1, frame=frame))
        st, h, p = rm.read(1)
        eh, ep, _ = oracle.protein_minimizer_nt(q, 9, 5, 1, frame)
        assert (st & L.ST_CODE_MASK) == L.ST_OK and np.array_equal(h, eh) and np.array_equal(p & L.POS_MASK, ep), frame
        eh0, ep0, _ = oracle.protein_minimizer_nt(small.tobytes().decode(), 9, 5, 1, frame)
        st0, h0, p0 = rm.read(0)
        assert np.array_equal(h0, eh0) and np.array_equal(p0 & L.POS_MASK, ep0)
        rm.close()
    rh = engine.run(b, engine.params(L.PROT_HASH, 10, codon_table=11, frame=-1))
    assert np.array_equal(rh.read(1)[1], oracle.protein_hashes_nt(q, 10, 11, -1))
    rh.close()
    pb = b.translate(1, 3)  # the stand-alone Translate
    assert pb.info()["n_bases"] == (n - 2) // 3 + (len(small) - 2) // 3
    pb.close()
    b.close()


def test_two_strand_kmer_codes_tile(engine, oracle, tiny_tiles):
    """NextKmer's two-strand mode (iterator.go:713-723) over tiles: forward codes per tile, then the second strand per sequence --
    including sequences whose 2(L-k+1) values no longer fit a 24-bit count, an illegal base (no second strand: the error comes
    first, iterator.go:746-748), and IUPAC letters (RevComInplace pairs letters, base2bit maps the pair)."""
    from bio_amd import sketches as S
    n = (1 << 23) + 5000
    rng = np.random.default_rng(4)
    big = np.array(list(b"ACGT"), np.uint8)[rng.integers(0, 4, n)]
    b = engine.batch_from_arrays(big, np.array([0, n], np.uint64))
    res = engine.run(b, engine.params(L.KMER, 21, canonical=False))
    assert res.info()["n_tuples"] == 2 * (n - 20)
    assert np.array_equal(res.read(0)[1], oracle.kmer_codes(big.tobytes().decode(), 21, False, False))
    os.environ["BSK_NO_TILES"] = "1"  # the per-lane kernels keep a sequence's count in 24 bits: refused, not wrapped
    try:
        with pytest.raises(S.DeviceError, match="2\\^23"):
            engine.run(b, engine.params(L.KMER, 21, canonical=False))
    finally:
        del os.environ["BSK_NO_TILES"]
    b.close()
    tiny_tiles(40, 16)
    r = random.Random(12)
    k = 9
    seqs = [rand_seq(r, 700), rand_seq(r, 300, "ACGTNRYKMacgtn"), rand_seq(r, 5), rand_seq(r, 9), rand_seq(r, 1200, "ACGTSWBDHVU")]
    bad = list(rand_seq(r, 800))
    bad[333] = "X"
    seqs.append("".join(bad))
    for circular in (False, True):
        b = engine.batch(seqs)
        res = engine.run(b, engine.params(L.KMER, k, canonical=False, circular=circular))
        for i, q in enumerate(seqs):
            st, h, _ = res.read(i)
            try:
                e = oracle.kmer_codes(q, k, False, circular)
            except oracle.OracleError as err:
                if err.name == "ErrShortSeq":
                    assert (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0
                else:
                    assert err.name == "ErrIllegalBase" and (st & L.ST_CODE_MASK) == L.ST_ILLEGAL and len(h) == 333 - k + 1
                    assert np.array_equal(h, oracle.kmer_codes(q[:333], k, False, False)[: 333 - k + 1])
                continue
            assert (st & L.ST_CODE_MASK) == L.ST_OK and np.array_equal(h, e), (i, circular, len(h), len(e))
        b.close()


@pytest.mark.parametrize("mode", ["deferred", "forced fallback", "round trips", "general kernels"])
def test_tiled_path_without_host_round_trips(engine, oracle, monkeypatch, mode):
    """Round 6: a tiled call sizes every array from a host-side bound of the tile count (n_bases / tp + n), launches the tile kernels once
    into slabs sized by the plan and synchronises ONCE, at its end -- that synchronisation also brings the overflow flags, and a call that
    finds one set runs again with the sizing run of rounds 2-5 (BSK_TEST_OVERFLOW=8 pretends one).  All three ways give the same result:
    every sequence against the oracle for minimizers and syncmers, digests for the ntHash stream; sequences of very different lengths,
    among them short ones without a tile and one below the length rule (the bound's empty tiles sit behind them)."""
    if mode == "forced fallback":
        monkeypatch.setenv("BSK_TEST_OVERFLOW", "8")
    elif mode == "round trips":
        monkeypatch.setenv("BSK_NO_TILE_DEFER", "1")
    elif mode == "general kernels":
        # the dense look-back kernels size their output by an ESTIMATE; the homopolymer and the repeat below select far more than it, so the first
        # launch overflows and (before the fix) its reference words pointed past the arrays -- which the deferred path handed to the stitch pass:
        # a memory fault (scripts/fuzz_campaign.py 21000000 12000, round 6).  Such plans keep the sizing loop now.
        monkeypatch.setenv("BSK_FORCE_GENERIC", "1")
    rng = random.Random(66)
    seqs = [rand_seq(rng, n) for n in (5000, 31, 12000, 20, 4097, 150, 65000, 8000, 40, 300000, 9999)]
    seqs.append("A" * 7000)
    seqs.append("ACGTTGCA" * 900)
    if mode == "general kernels":
        seqs += ["C" * 30000, "GT" * 20000, "A" * 50000]
    b = engine.batch(seqs)
    for kind, pk, fn in ((L.MINIMIZER, dict(k=21, w=11), lambda q: oracle.minimizer(q, 21, 11, False, closed=True)),
                         (L.SYNCMER, dict(k=31, s=11), lambda q: oracle.syncmer(q, 31, 11, False, closed=True))):
        res = engine.run(b, engine.params(kind, **pk))
        assert "over tiles" in res.plan()["kernel"], res.plan()
        for i, q in enumerate(seqs):
            st, h, p = res.read(i)
            try:
                eh, ep, es, fl = fn(q)
            except oracle.OracleError as e:
                assert e.name == "ErrShortSeq" and (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0, (i, len(q))
                continue
            assert (st & L.ST_CODE_MASK) == L.ST_OK and np.array_equal(h, eh), (mode, kind, i, len(q))
            assert np.array_equal(p & L.POS_MASK, ep) and np.array_equal(p >> 31, es), (mode, kind, i, len(q))
        res.close()
    res = engine.run(b, engine.params(L.NTHASH, 21))
    d = res.digest()
    data = np.frombuffer("".join(seqs).encode(), np.uint8)
    offs = np.zeros(len(seqs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(q) for q in seqs])
    nt, ck = oracle.batch_run(2, data, offs, 21, 0, threads=1)
    assert d["n_tuples"] == nt and d["checksum"] == ck, mode
    res.close()
    b.close()


@pytest.mark.parametrize("k,w", [(21, 11), (31, 13), (15, 4), (21, 5), (64, 8), (33, 7), (11, 9), (25, 12)])
def test_minimizers_of_long_sequences_as_dense_tiles(engine, oracle, monkeypatch, k, w):
    """Round 6 (kernels_minimizer_pf.hpp): the tile kernel k_minimizer_pft writes the FINAL tuples -- only the positions a tile owns, shifted
    to the sequence, every unit packed behind the one before -- and no stitch pass runs.  Every sequence against the oracle's closed form
    (hashes, positions, strands, first-window flag); homopolymers, short repeats and a k-mer with its reverse complement inside one window
    are key ties of the packed machine: those tiles go through the in-kernel exact path.  Same digest as the slab + stitch path."""
    rng = random.Random(100 * k + w)
    seqs = [rand_seq(rng, n) for n in (5000, k + w - 2, 12000, 20, 4097, 150, 40000, 8000, k + w - 1, 90000, 9999, 333)]
    seqs.append("A" * 3000 + rand_seq(rng, 4000))                   # a homopolymer run inside an ordinary sequence
    seqs.append(rand_seq(rng, 2500) + "ACGTTGCA" * 200 + rand_seq(rng, 2500))
    pal = rand_seq(rng, k)
    rc = pal[::-1].translate(str.maketrans("ACGT", "TGCA"))
    seqs.append(rand_seq(rng, 3000) + pal + "AC" + rc + rand_seq(rng, 3000))  # equal canonical hashes a few positions apart
    seqs.append("AAAAA" + rand_seq(rng, 6000))                         # (first-window region with repeated k-mers when k is small)
    monkeypatch.setenv("BSK_TILE_DENSE", "1")
    b = engine.batch(seqs)
    p = engine.params(L.MINIMIZER, k, w=w)
    res = engine.run(b, p)
    assert "k_minimizer_pft<%d>" % w in res.plan()["kernel"] and "over tiles" in res.plan()["kernel"], res.plan()
    for i, q in enumerate(seqs):
        st, h, pos = res.read(i)
        try:
            eh, ep, es, fl = oracle.minimizer(q, k, w, False, closed=True)
        except oracle.OracleError as e:
            assert e.name == "ErrShortSeq" and (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0, (i, len(q))
            continue
        assert (st & L.ST_CODE_MASK) == L.ST_OK and len(h) == len(eh), (k, w, i, len(q), len(h), len(eh))
        assert np.array_equal(h, eh) and np.array_equal(pos & L.POS_MASK, ep) and np.array_equal(pos >> 31, es), (k, w, i, len(q))
        assert (st & 0xF0) == fl, (k, w, i, st, fl)
    d = res.digest()
    res.close()
    monkeypatch.delenv("BSK_TILE_DENSE")
    res = engine.run(b, p)
    assert "k_minimizer_pft" not in res.plan()["kernel"], res.plan()
    assert res.digest() == d
    res.close()
    b.close()


def test_dense_tiles_fall_back_when_the_batch_selects_more_than_expected(engine, oracle, monkeypatch):
    """The dense path sizes the sequence result by the expected density (2 / (w + 1) per position + a quarter); a batch of homopolymers
    selects EVERY position: the kernel raises its flag instead of writing past the arrays and the call runs again through slabs + stitch."""
    seqs = ["A" * 2000000, "C" * 150000, "ACGT" * 3, "GT" * 60000]
    monkeypatch.setenv("BSK_TILE_DENSE", "1")
    b = engine.batch(seqs)
    res = engine.run(b, engine.params(L.MINIMIZER, 21, w=11))
    assert "over tiles" in res.plan()["kernel"] and "k_minimizer_pft" not in res.plan()["kernel"], res.plan()
    for i, q in enumerate(seqs):
        st, h, pos = res.read(i)
        try:
            eh, ep, es, fl = oracle.minimizer(q, 21, 11, False, closed=True)
        except oracle.OracleError:
            assert (st & L.ST_CODE_MASK) == L.ST_SHORT
            continue
        assert np.array_equal(h, eh) and np.array_equal(pos & L.POS_MASK, ep), i
    res.close()
    b.close()
