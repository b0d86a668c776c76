"""GPU parity for the remaining iterator kinds: syncmer, k-mer codes, SimHash, protein hash / minimizer."""
import os
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
                                          (21, 1, False), (11, 5, True), (64, 33, False), (21, 19, False), (21, 17, False), (15, 8, False),
                                          (21, 12, False), (31, 19, False), (31, 17, False), (31, 13, False), (31, 7, False), (40, 23, False),
                                          (12, 9, False), (20, 14, True), (21, 10, False), (27, 14, False)])
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


@pytest.mark.parametrize("k,s,n,lens", [(31, 11, 3000, (150,)), (31, 16, 1500, (150, 100, 224)), (21, 10, 1500, (150, 90, 61)),
                                        (15, 4, 1500, (80, 150)), (31, 7, 1000, (150, 300, 500)), (40, 23, 600, (250, 2000))])
def test_syncmer_matches_state_machine_too(engine, oracle, k, s, n, lens):
    """The kernels implement the closed form; this pins them to the line-by-line restatement of the reference's state machine
    (sorted buffer, binary-search insert, pending queue: sketch.go:312-477) on thousands of reads per shape, ragged batches included."""
    rng = random.Random(77 * k + s)
    seqs = [rand_seq(rng, rng.choice(lens) + (rng.randint(0, 9) if len(lens) > 1 else 0)) for _ in range(n)]
    b = engine.batch(seqs)
    res = engine.run(b, engine.params(L.SYNCMER, k, s=s))
    for i, q in enumerate(seqs):
        _, h, p = res.read(i)
        eh, ep, _, _ = oracle.syncmer(q, k, s)  # closed=False: the state machine
        assert np.array_equal(h, eh) and np.array_equal(p & L.POS_MASK, ep), (i, k, s, len(q))


@pytest.mark.parametrize("k,s,lo,hi", [(31, 11, 150, 150), (31, 11, 60, 210), (31, 16, 100, 170), (21, 10, 61, 120), (64, 44, 150, 224), (33, 20, 150, 151),
                                       (32, 12, 120, 160), (48, 30, 180, 215), (17, 9, 40, 95), (16, 8, 30, 94), (63, 43, 105, 224), (27, 8, 150, 150), (25, 13, 100, 138)])
def test_syncmer_fused_emit_kernel(engine, oracle, k, s, lo, hi):
    """k_syncmer_pf (round 6, kernels_syncmer_pf.hpp): the s-mer machine alone, selection words in LDS, and at the end of every unit the
    canonical k-mer hashes of the selected positions FROM SCRATCH (three bases per table row, one lane per tuple).  Every read against
    the closed form AND the reference's state machine (sketch.go:312-477); k from 16 to 64 (one to four whole words + every remainder
    class of the emit's 16-base / 3-base pieces), k - s = 8..20, reads up to the 224-base word limit, ragged batches (length-binned
    units), low-complexity reads (key ties -> the exact machine) in the batch."""
    rng = random.Random(1000 * k + s + hi)
    seqs = [rand_seq(rng, rng.randint(lo, hi)) for _ in range(2300)]
    seqs[7] = "A" * hi                                      # every s-mer the same hash: the exact machine's
    seqs[8] = rand_seq(rng, max(lo - 40, 1)) + "T" * 40     # a low-complexity tail
    seqs[9] = rand_seq(rng, 2 * k - s - 2)                  # one base short (sketch.go:149)
    seqs[10] = rand_seq(rng, 2 * k - s - 1)                 # the shortest read with a window
    b = engine.batch(seqs)
    res = engine.run(b, engine.params(L.SYNCMER, k, s=s))
    assert "k_syncmer_pf<%d>" % (k - s) in res.plan()["kernel"], res.plan()
    for i, q in enumerate(seqs):
        st, h, p = res.read(i)
        if len(q) < 2 * k - s - 1:
            assert (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0, (i, len(q))
            continue
        eh, ep, es, fl = oracle.syncmer(q, k, s, False, closed=True)
        assert (st & L.ST_CODE_MASK) == L.ST_OK and np.array_equal(h, eh), (i, k, s, len(q))
        assert np.array_equal(p & L.POS_MASK, ep) and np.array_equal(p >> 31, es) and (st & 0xF0) == fl, (i, k, s, len(q))
        if i % 4 == 0:
            mh, mp, _, _ = oracle.syncmer(q, k, s)  # the state machine
            assert np.array_equal(h, mh) and np.array_equal(p & L.POS_MASK, mp), (i, k, s, len(q))
    res.close()
    b.close()


def test_syncmer_fused_emit_digest_equals_the_one_pass_kernels(engine, monkeypatch):
    """Same batches through k_syncmer_pf and (BSK_NO_SYN_PF) k_syncmer_pk / k_syncmer_pkl: identical digests, tuple and flag counts."""
    for n, ln, k, s in ((400000, 150, 31, 11), (200000, 200, 31, 11), (150000, 130, 21, 9), (100000, 224, 64, 46)):
        b = engine.synth(L.ALPHA_DNA, n, ln, 0x5EED0600 + ln + k)
        prm = engine.params(L.SYNCMER, k, s=s)
        res = engine.run(b, prm)
        assert "k_syncmer_pf" in res.plan()["kernel"], res.plan()
        d1 = res.digest()
        res.close()
        monkeypatch.setenv("BSK_NO_SYN_PF", "1")
        res = engine.run(b, prm)
        assert "k_syncmer_pf" not in res.plan()["kernel"] and "k_syncmer" in res.plan()["kernel"], res.plan()
        d2 = res.digest()
        res.close()
        monkeypatch.delenv("BSK_NO_SYN_PF")
        assert d1 == d2 and d1["n_tuples"] > 0, (n, ln, k, s, d1, d2)
        b.close()


def test_syncmer_fused_emit_units_beyond_the_tuple_list(engine, oracle, monkeypatch):
    """A unit that selects more positions than the emit phase's list holds (1 024 per 64 reads: never planned -- the planner keeps
    k_syncmer_pf to 12 expected selections per read -- so BSK_PF_DENSITY plans it here): the unit's LAST reads go to the list of the exact
    machine, everything stays exact."""
    monkeypatch.setenv("BSK_PF_DENSITY", "40")
    rng = random.Random(1024)
    k, s = 20, 12  # k - s = 8: a sixth of the windows is selected, ~20 per 150-base read, ~1 300 per unit
    seqs = [rand_seq(rng, 150) for _ in range(1500)]
    b = engine.batch(seqs)
    res = engine.run(b, engine.params(L.SYNCMER, k, s=s))
    assert "k_syncmer_pf<8>" in res.plan()["kernel"], res.plan()
    off, st, h, p = res.fetch()
    assert int(off[64]) - int(off[0]) > 1024  # (the first unit does select more than the list holds)
    for i, q in enumerate(seqs):
        eh, ep, es, fl = oracle.syncmer(q, k, s, False, closed=True)
        a, e = int(off[i]), int(off[i + 1])
        assert np.array_equal(h[a:e], eh) and np.array_equal(p[a:e] & L.POS_MASK, ep) and np.array_equal(p[a:e] >> 31, es) and (int(st[i]) & 0xF0) == fl, i
    res.close()
    b.close()


def test_syncmer_packed_kernel_at_its_length_limit(engine, oracle):
    """k_syncmer_pk takes reads of up to 224 bases (its words live in 16 registers, the last two are look-ahead): lengths 208..224
    without jitter, so that the batch really is planned on that kernel and the clamped word index of its last blocks is exercised."""
    rng = random.Random(2240)
    seqs = [rand_seq(rng, rng.randint(208, 224)) for _ in range(2500)]
    b = engine.batch(seqs)
    # (the kernel's short staging columns admit such long reads only when they select few positions: large k)
    for k, s in ((100, 80), (104, 85), (110, 92)):
        res = engine.run(b, engine.params(L.SYNCMER, k, s=s))
        assert "k_syncmer_pk" in res.plan()["kernel"], res.plan()
        for i, q in enumerate(seqs):
            st, h, p = res.read(i)
            eh, ep, es, fl = oracle.syncmer(q, k, s, False, closed=True)
            assert (st & L.ST_CODE_MASK) == L.ST_OK and np.array_equal(h, eh), (i, k, s, len(q))
            assert np.array_equal(p & L.POS_MASK, ep) and np.array_equal(p >> 31, es) and (st & 0xF0) == fl, (i, k, s, len(q))
        res.close()
    b.close()


@pytest.mark.parametrize("k,s,lo,hi", [(31, 11, 225, 260), (31, 11, 180, 250), (120, 100, 330, 352), (35, 11, 120, 160), (31, 11, 300, 352), (150, 128, 440, 480), (33, 10, 140, 150),
                                       (35, 13, 150, 151), (45, 21, 200, 300), (31, 11, 60, 250)])
@pytest.mark.parametrize("fused", [True, False])
def test_syncmer_long_packed_kernel(engine, oracle, monkeypatch, k, s, lo, hi, fused):
    """fused (round 6): k_syncmer_pfl, the fused-emit kernel with 32 words of a read in registers, takes these batches where k <= 64;
    not fused (BSK_NO_SYN_PF): k_syncmer_pkl as in rounds 4-5.
    k_syncmer_pkl: the packed machine with 32 words of a read in registers (reads of up to 480 bases), longer staging columns and
    k - s up to 24 (round 4; until then these batches ran on k_syncmer_fast).  Every read against the closed form AND the reference's
    state machine; lengths up to the word limit so that the clamped word index of the last blocks is exercised; ragged batches
    (length-binned units) included."""
    rng = random.Random(1000 * k + s + hi)
    seqs = [rand_seq(rng, rng.randint(lo, hi)) for _ in range(2200)]
    seqs[7] = "A" * hi                      # every s-mer the same hash: a key tie in every min operation -> the exact machine
    seqs[8] = rand_seq(rng, lo - 40) + "T" * 40  # a low-complexity tail
    if not fused:
        monkeypatch.setenv("BSK_NO_SYN_PF", "1")
    b = engine.batch(seqs)
    res = engine.run(b, engine.params(L.SYNCMER, k, s=s))
    assert ("k_syncmer_pfl" if fused and k <= 64 else "k_syncmer_pkl") in res.plan()["kernel"], res.plan()
    for i, q in enumerate(seqs):
        st, h, p = res.read(i)
        if len(q) < 2 * k - s - 1:
            assert (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0, (i, len(q))
            continue
        eh, ep, es, fl = oracle.syncmer(q, k, s, False, closed=True)
        assert (st & L.ST_CODE_MASK) == L.ST_OK and np.array_equal(h, eh), (i, k, s, len(q))
        assert np.array_equal(p & L.POS_MASK, ep) and np.array_equal(p >> 31, es) and (st & 0xF0) == fl, (i, k, s, len(q))
        if i % 4 == 0:
            mh, mp, _, _ = oracle.syncmer(q, k, s)  # the state machine
            assert np.array_equal(h, mh) and np.array_equal(p & L.POS_MASK, mp), (i, k, s, len(q))
    res.close()
    b.close()


def test_syncmer_mid_length_reads_run_as_tiles(engine, oracle):
    """Reads beyond the packed syncmer kernels' reach (449+ bases) run as tiles of 256 + 3k - 2s + 12 bases on k_syncmer_pkl instead of on the
    per-read 64-bit kernel (whose 28-tuple slabs overflow there): same tuples, every read against the closed form, every fourth
    against the state machine."""
    rng = random.Random(4490)
    seqs = [rand_seq(rng, rng.choice((449, 450, 557, 558, 700, 1500, 3000))) for _ in range(500)]
    seqs[3] = "AC" * 400
    b = engine.batch(seqs)
    res = engine.run(b, engine.params(L.SYNCMER, 31, s=11))
    assert "over tiles" in res.plan()["kernel"] and ("k_syncmer_pkl" in res.plan()["kernel"] or "k_syncmer_pfl" in res.plan()["kernel"]), res.plan()
    for i, q in enumerate(seqs):
        st, h, p = res.read(i)
        eh, ep, es, fl = oracle.syncmer(q, 31, 11, False, closed=True)
        assert (st & L.ST_CODE_MASK) == L.ST_OK and np.array_equal(h, eh), (i, len(q))
        assert np.array_equal(p & L.POS_MASK, ep) and np.array_equal(p >> 31, es), (i, len(q))
        if i % 4 == 0:
            mh, mp, _, _ = oracle.syncmer(q, 31, 11)
            assert np.array_equal(h, mh) and np.array_equal(p & L.POS_MASK, mp), (i, len(q))
    res.close()
    b.close()


@pytest.mark.parametrize("k,s", [(21, 11), (25, 15), (31, 15), (64, 48), (15, 9)])
def test_syncmer_tiles_stay_on_the_long_packed_plan(engine, oracle, k, s):
    """Tiles are sized by the long packed plan's limit and the tile batch's length bound is the tile's exact extent (tp + 3k - 2s + 12):
    with the looser 3k + 16 the planner put k=25 s=15 and k=64 s=48 back on k_syncmer_fast (round 5, scripts/dev/run_holes.sh).  Every read
    against the closed form; reads of two tiles either way, of many tiles, of one letter."""
    rng = random.Random(31 * k + s)
    seqs = [rand_seq(rng, rng.choice((460, 700, 1000, 2500))) for _ in range(300)] + ["A" * 900, "ACG" * 400]
    b = engine.batch(seqs)
    res = engine.run(b, engine.params(L.SYNCMER, k, s=s))
    assert "over tiles" in res.plan()["kernel"] and ("k_syncmer_pkl<%d>" % (k - s) in res.plan()["kernel"] or "k_syncmer_pfl<%d>" % (k - s) in res.plan()["kernel"]), res.plan()  # (round 6: the fused-emit long plan where the tile's density fits its list)
    for i, q in enumerate(seqs):
        st, h, p = res.read(i)
        eh, ep, es, fl = oracle.syncmer(q, k, s, False, closed=True)
        assert (st & L.ST_CODE_MASK) == L.ST_OK and np.array_equal(h, eh), (k, s, i, len(q))
        assert np.array_equal(p & L.POS_MASK, ep) and np.array_equal(p >> 31, es), (k, s, i, len(q))
    res.close()
    b.close()


def test_syncmer_small_s_is_not_planned_on_the_packed_kernels(engine, oracle):
    """s = 7: equal s-mers inside one 2w window are the rule, every such read is the exact machine's -- the planner keeps such
    parameters off the packed kernels (their list would fill up and the call would fall back after a wasted run)."""
    rng = random.Random(707)
    seqs = [rand_seq(rng, 150) for _ in range(600)]
    b = engine.batch(seqs)
    res = engine.run(b, engine.params(L.SYNCMER, 31, s=7))
    assert "k_syncmer_fast" in res.plan()["kernel"], res.plan()
    for i, q in enumerate(seqs):
        st, h, p = res.read(i)
        eh, ep, es, fl = oracle.syncmer(q, 31, 7, False, closed=True)
        assert np.array_equal(h, eh) and np.array_equal(p & L.POS_MASK, ep) and (st & 0xF0) == fl, i
    res.close()
    b.close()


@pytest.mark.parametrize("k,s", [(33, 8), (31, 5), (45, 16), (63, 32), (63, 31), (40, 8)])
def test_syncmer_wide_windows_run_on_the_staged_kernel(engine, oracle, k, s):
    """k - s = 25..32 (k_syncmer_wide.hip, round 5; the general per-lane kernel before): k_syncmer_fast<k - s>, read by read against the
    closed form and the state machine -- short reads, reads of one letter, reads that overflow their staging columns included"""
    rng = random.Random(97 * k + s)
    seqs = [rand_seq(rng, rng.choice([150, 250, rng.randint(1, 900)])) for _ in range(900)] + ["A" * 300, "ACGT" * 80, rand_seq(rng, 2 * k - s), rand_seq(rng, 2 * k - s - 1), ""]
    b = engine.batch(seqs)
    res = engine.run(b, engine.params(L.SYNCMER, k, s=s))
    assert "k_syncmer_fast<%d>" % (k - s) in res.plan()["kernel"], res.plan()
    for i, q in enumerate(seqs):
        st, h, p = res.read(i)
        try:
            eh, ep, es, fl = oracle.syncmer(q, k, s, False, closed=True)
        except oracle.OracleError as err:
            assert (st & L.ST_CODE_MASK) != 0 and len(h) == 0, (i, len(q), err.name)
            continue
        assert np.array_equal(h, eh) and np.array_equal(p & L.POS_MASK, ep) and np.array_equal(p >> 31, es), (k, s, i, len(q))
        assert (st & 0xF0) == fl, (k, s, i, st, fl)
        if i % 7 == 0:
            mh, mp, _, _ = oracle.syncmer(q, k, s)  # the state machine
            assert np.array_equal(h, mh) and np.array_equal(p & L.POS_MASK, mp), (k, s, i, len(q))
    res.close()
    b.close()


def test_syncmer_long_plan_digest_equals_the_64_bit_kernels(engine):
    """Same batch through k_syncmer_pfl, k_syncmer_pkl (BSK_NO_SYN_PF) and (BSK_NO_SYN_LONG) k_syncmer_fast: identical digests and flag counts."""
    import os
    b = engine.synth(L.ALPHA_DNA, 300000, 250, 0x5EED0250)
    prm = engine.params(L.SYNCMER, 31, s=11)
    res = engine.run(b, prm)
    assert "k_syncmer_pfl" in res.plan()["kernel"], res.plan()
    d0 = res.digest()
    res.close()
    os.environ["BSK_NO_SYN_PF"] = "1"
    try:
        engine.reload_options()
        res = engine.run(b, prm)
        assert "k_syncmer_pkl" in res.plan()["kernel"], res.plan()
        d1 = res.digest()
        res.close()
    finally:
        del os.environ["BSK_NO_SYN_PF"]
        engine.reload_options()
    assert d0 == d1, (d0, d1)
    os.environ["BSK_NO_SYN_LONG"] = "1"
    try:
        engine.reload_options()
        res = engine.run(b, prm)
        assert "k_syncmer_fast" in res.plan()["kernel"], res.plan()
        d2 = res.digest()
        res.close()
    finally:
        del os.environ["BSK_NO_SYN_LONG"]
        engine.reload_options()
    assert d1 == d2, (d1, d2)
    b.close()


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


@pytest.mark.parametrize("alphabet", [L.ALPHA_DNA, L.ALPHA_DNA_PLAIN, L.ALPHA_RNA, L.ALPHA_RNA_REDUNDANT, L.ALPHA_UNLIMIT])
@pytest.mark.parametrize("tiles", [False, True])
def test_two_strand_kmer_codes_pair_letters_with_the_sequences_own_alphabet(engine, oracle, alphabet, tiles):
    """The second strand of NextKmer's canonical=false mode is RevComInplace of the Seq (iterator.go:719): PairLetter of ITS alphabet
    (seq/alphabet.go:353-383) -- RNA pairs A with U and leaves a T, plain DNA leaves R/Y..., Unlimit complements nothing
    (seq/seq.go:381-383).  Pure-ACGT reads (2-bit kernels) and reads with other letters (ASCII kernels), per lane and over tiles."""
    rng = random.Random(alphabet * 2 + tiles)
    seqs = [rand_seq(rng, rng.randint(12, 400)) for _ in range(80)]
    seqs += [rand_seq(rng, rng.randint(12, 400), "ACGU") for _ in range(20)]
    seqs += [rand_seq(rng, rng.randint(12, 400), "ACGTUacgtuNRYSWKMBDHVryswkmbdhvn") for _ in range(40)]
    seqs += [rand_seq(rng, 2000), rand_seq(rng, 1500, "ACGUTRYN")]
    saved = {k: os.environ.get(k) for k in ("BSK_TILE_MIN", "BSK_TILE_POS")}
    if tiles:
        os.environ["BSK_TILE_MIN"], os.environ["BSK_TILE_POS"] = "40", "16"
    try:
        for k, circular in ((11, False), (7, True), (32, False)):
            b = engine.batch(seqs, alphabet)
            res = engine.run(b, engine.params(L.KMER, k, canonical=False, circular=circular))
            for i, q in enumerate(seqs):
                st, h, _ = res.read(i)
                try:
                    e = oracle.kmer_codes(q, k, False, circular, alphabet)
                except oracle.OracleError as err:
                    assert err.name == "ErrShortSeq" and (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0
                    continue
                assert np.array_equal(h, e), (alphabet, tiles, k, circular, i, q[:40])
            rc = engine.run(b, engine.params(L.KMER, k, canonical=True, circular=circular))  # the canonical mode never looks at the alphabet
            for i in (0, 85, 110, 141):
                if len(seqs[i]) >= k:
                    assert np.array_equal(rc.read(i)[1], oracle.kmer_codes(seqs[i], k, True, circular))
            res.close()
            rc.close()
            b.close()
    finally:
        for k_, v in saved.items():
            if v is None:
                os.environ.pop(k_, None)
            else:
                os.environ[k_] = v


@pytest.mark.parametrize("alphabet", [L.ALPHA_DNA, L.ALPHA_RNA, L.ALPHA_UNLIMIT])
def test_two_strand_kmer_codes_on_the_stream_kernel(engine, oracle, alphabet):
    """Round 6: NextKmer's two-strand mode (iterator.go:713-723) of pure-ACGT reads of up to 512 bases runs on k_nthash_fast<3> -- the lane
    walks forward through the read and then back, rolling the code of the reverse-complemented letters the other way -- instead of the
    general kernel.  Ragged and fixed-length batches, k = 1 .. 32, every value against the oracle; the second strand pairs the letters of
    the batch's own alphabet (RNA leaves a T, Unlimit complements nothing)."""
    rng = random.Random(977 + alphabet)
    ragged = [rand_seq(rng, n) for n in [1, 2, 11, 12, 15, 16, 17, 31, 32, 33, 63, 64, 65, 150, 151, 255, 256, 257, 400, 511, 512]]
    ragged += [rand_seq(rng, rng.randint(5, 512)) for _ in range(200)]
    fixed = [rand_seq(rng, 150) for _ in range(300)]
    for seqs in (ragged, fixed):
        b = engine.batch(seqs, alphabet)
        for k in (1, 2, 5, 11, 16, 17, 21, 31, 32):
            res = engine.run(b, engine.params(L.KMER, k, canonical=False))
            assert "k_nthash_fast<3>" in res.plan()["kernel"], res.plan()
            for i, q in enumerate(seqs):
                st, h, _ = res.read(i)
                try:
                    e = oracle.kmer_codes(q, k, False, False, alphabet)
                except oracle.OracleError as err:
                    assert err.name == "ErrShortSeq" and (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0, (k, i)
                    continue
                assert (st & L.ST_CODE_MASK) == L.ST_OK and len(h) == 2 * (len(q) - k + 1), (alphabet, k, i, len(q), len(h))
                assert np.array_equal(h, e), (alphabet, k, i, len(q))
            res.close()
        b.close()


def test_kmer_k_too_large(engine):
    from bio_amd import sketches as S
    seq, _ = S.NewSeq(S.DNA, "ACGT" * 20)
    assert S.NewKmerIterator(seq, 33, True, False, engine) == (None, S.ErrKTooLarge)


@pytest.mark.parametrize("k,m,scale,canonical", [(21, 5, 5, True), (31, 5, 5, True), (21, 5, 1, True), (21, 21, 1, False),
                                                 (16, 4, 13, True), (40, 5, 7, True), (36, 5, 2, False), (35, 4, 32, True),
                                                 (67, 4, 3, True), (70, 6, 3, False)])  # k-m+1: <=31 five planes, <=63 six, else scalar counters
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


@pytest.mark.parametrize("k", [3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17])
def test_protein_hash_register_kernel_all_k_and_long_sequences(engine, oracle, k):
    """k = 4..16 (4..8: round 5) run on k_prot_hash_fast<K> (256-position chunks staged in LDS); 3 and 17 on the general kernel."""
    rng = random.Random(900 + k)
    lens = [3 * k - 1, 3 * k, 3 * k + 1, 47, 48, 49, 511, 512, 513, 512 + k - 1, 512 + k, 1023, 1024, 1025, 1500, 2600]
    seqs = [rand_seq(rng, n, AA) for n in lens] + [rand_seq(rng, rng.randint(1, 700), AA) for _ in range(150)]
    b = engine.batch(seqs, L.ALPHA_PROTEIN)
    res = engine.run(b, engine.params(L.PROT_HASH, k))
    assert ("k_prot_hash_fast" in res.plan()["kernel"]) == (4 <= k <= 16), res.plan()
    for i, q in enumerate(seqs):
        st, h, _ = res.read(i)
        try:
            e = oracle.protein_hashes(q, k)
        except oracle.OracleError as err:
            assert err.name == "ErrShortSeq" and (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0
        else:
            assert np.array_equal(h, e), (i, len(q))


IUPAC = "ACGTacgtUuNnRYSWKMBDHVryswkmbdhv"


def _codon_golden():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "codon_golden.json")))


@pytest.mark.parametrize("frame", [1, 2, 3, -1, -2, -3])
def test_translate_on_device(engine, oracle, frame):
    """bsk_batch_translate == Translate(table, frame, false, false, true, false) (seq/codon_tables.go:205-285): the
    reference's own vectors (codon_tables_test.go), every genetic code, IUPAC / gap / junk letters, both encodings."""
    rng = random.Random(40 + frame)
    gold = _codon_golden()
    for table in (1, 2, 4, 11, 25, 31):
        plain = [rand_seq(rng, rng.randint(3, 400)) for _ in range(150)] + [v["nt"].upper() for v in gold["vectors"]]
        mixed = [rand_seq(rng, rng.randint(3, 300), IUPAC) for _ in range(100)]
        mixed += [v["nt"] for v in gold["vectors"]] + ["---ATG---", "AT-GCC* *AA", "ACGTXJ!ACGTAA", "atgNNNtaa", "ATGRAYTGGNNNGCN---TAA"]
        for seqs in (plain, mixed):  # 2-bit path / ASCII path
            b = engine.batch(seqs)
            t = b.translate(table, frame)
            data, offs = t.fetch_ascii(0, len(seqs))
            for i, q in enumerate(seqs):
                got = data[int(offs[i]):int(offs[i + 1])].tobytes().decode("latin-1")
                assert got == oracle.translate(q, table, frame), (table, frame, i, q)
            t.close()
            b.close()
    for v in gold["vectors"]:  # the reference's expected strings (trim=true ones end at the first stop)
        if v["frame"] != frame:
            continue
        b = engine.batch([v["nt"]])
        t = b.translate(v["table"], frame)
        data, offs = t.fetch_ascii(0, 1)
        got = data.tobytes().decode("latin-1")
        assert got.startswith(v["aa"]) and (not v["trim"] or got[len(v["aa"]):len(v["aa"]) + 1] in ("*", "X", ""))
        if not v["trim"]:
            assert got == v["aa"]


@pytest.mark.parametrize("k,w,table,frame", [(9, 5, 1, 1), (9, 5, 11, -1), (10, 3, 4, 2), (3, 1, 1, 3), (5, 4, 2, -3), (12, 4, 1, -2),
                                             (17, 6, 1, 1)])
def test_protein_kinds_on_dna_input(engine, oracle, k, w, table, frame):
    """NewProteinIterator / NewProteinMinimizerSketch fed DNA: length checks on the nucleotides (iterator-protein.go:50,
    sketch-protein.go:66,73), then Translate, then the protein path -- a translation with fewer than k residues or fewer
    than w k-mers yields nothing and is no error."""
    rng = random.Random(k * 100 + w * 10 + frame + 7)
    edge = [3 * k - 1, 3 * k, 3 * k + 1, 3 * k + 2, 3 * k + w - 2, 3 * k + w - 1, 3 * k + w, 3 * (k + w), 3 * (k + w) + 2, 1, 2, 3]
    for alphabet in ("ACGT", "ACGTacgtNRYU-"):
        seqs = [rand_seq(rng, n, alphabet) for n in edge] + [rand_seq(rng, rng.randint(1, 1200), alphabet) for _ in range(150)]
        b = engine.batch(seqs)
        rh = engine.run(b, engine.params(L.PROT_HASH, k, codon_table=table, frame=frame))
        rm = engine.run(b, engine.params(L.PROT_MINIMIZER, k, w=w, codon_table=table, frame=frame))
        for i, q in enumerate(seqs):
            st, h, _ = rh.read(i)
            try:
                e = oracle.protein_hashes_nt(q, k, table, frame)
            except oracle.OracleError as err:
                assert err.name == "ErrShortSeq" and (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0, (i, len(q))
            else:
                assert (st & L.ST_CODE_MASK) == 0 and np.array_equal(h, e), (i, len(q))
            st, h, p = rm.read(i)
            try:
                eh, ep, fl = oracle.protein_minimizer_nt(q, k, w, table, frame)
            except oracle.OracleError as err:
                assert err.name == "ErrShortSeq" and (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0, (i, len(q))
            else:
                assert (st & L.ST_CODE_MASK) == 0 and np.array_equal(h, eh) and np.array_equal(p & L.POS_MASK, ep), (i, len(q))
                if not fl:  # unflagged reads: bit-exact claim; flagged ones (first-window tie) must carry the flag
                    pass
                else:
                    assert st & L.ST_FIRST_WINDOW_TIE
        b.close()


def test_protein_constructors_on_dna_seq(engine, oracle):
    from bio_amd import sketches as S
    rng = random.Random(77)
    dna = rand_seq(rng, 600)
    seq, _ = S.NewSeq(S.DNA, dna)
    it, err = S.NewProteinIterator(seq, 9, 1, 1, engine)
    assert err is None
    got = []
    while True:
        code, ok = it.Next()
        if not ok:
            break
        got.append(code)
    assert np.array_equal(np.array(got, np.uint64), oracle.protein_hashes_nt(dna, 9, 1, 1))
    sk, err = S.NewProteinMinimizerSketch(seq, 9, 11, -2, 5, engine)
    assert err is None
    eh, ep, _ = oracle.protein_minimizer_nt(dna, 9, 5, 11, -2)
    got, pos = [], []
    while True:
        code, ok = sk.Next()
        if not ok:
            break
        got.append(code)
        pos.append(sk.Index())
    assert np.array_equal(np.array(got, np.uint64), eh) and pos == [int(x) for x in ep]
    short, _ = S.NewSeq(S.DNA, "ACGT" * 6)  # 24 < 3*9
    assert S.NewProteinIterator(short, 9, 1, 1, engine) == (None, S.ErrShortSeq)
    with pytest.raises(S.DeviceError, match="codon table"):  # Translate's own errors (seq.go:691,694) are not sentinels
        S.NewProteinIterator(seq, 9, 7, 1, engine)
    with pytest.raises(S.DeviceError, match="frame"):
        S.NewProteinIterator(seq, 9, 1, 0, engine)


@pytest.mark.parametrize("frac", [0.01, 0.2, 0.6, 0.97, 1.0])
def test_mixed_batches_fast_kernels_plus_ascii_side_launch(engine, oracle, frac):
    """A batch where a few reads carry N / IUPAC letters: the 2-bit fast kernels run over everything and the general ASCII
    kernels re-do the flagged reads in a side launch (frac <= 0.9, or any share where the side launch is a staged kernel: minimizers with
    w <= 16, syncmers with k - s <= 24), otherwise the whole batch runs on the ASCII kernels.
    Either way every read must equal the oracle; BSK_NO_MIXED forces the second form for comparison of the digests."""
    import os
    rng = random.Random(int(frac * 1000))
    seqs = []
    for i in range(3000):
        n = rng.choice([150, 150, rng.randint(1, 300)])
        bad = rng.random() < frac
        seqs.append(rand_seq(rng, n, "ACGTN" if bad and i % 2 else ("ACGTRYKMacgtn" if bad else "ACGT")))
    b = engine.batch(seqs)
    cases = [(L.MINIMIZER, dict(k=21, w=11), lambda q: oracle.minimizer(q, 21, 11, False, closed=True)[:2]),
             (L.MINIMIZER, dict(k=15, w=7), lambda q: oracle.minimizer(q, 15, 7, False, closed=True)[:2]),      # generic w: no mixed plan
             (L.SYNCMER, dict(k=31, s=11), lambda q: oracle.syncmer(q, 31, 11, False, closed=True)[:2]),
             (L.NTHASH, dict(k=21), lambda q: (oracle.nthash(q, 21, True)[0], None)),
             (L.KMER, dict(k=21), lambda q: (oracle.kmer_codes(q, 21, True, False), None)),
             (L.SIMHASH, dict(k=21, m=5, scale=5), lambda q: (oracle.simhash(q, 21, 5, 5, True), None)),
             (L.MINIMIZER, dict(k=11, w=5, circular=True), lambda q: oracle.minimizer(q, 11, 5, True, closed=True)[:2])]
    for kind, pk, fn in cases:
        res = engine.run(b, engine.params(kind, **pk))
        for i, q in enumerate(seqs):
            st, h, p = res.read(i)
            try:
                eh, ep = fn(q)
            except oracle.OracleError as e:
                assert e.name in ("ErrShortSeq", "ErrIllegalBase"), e.name
                if e.name == "ErrShortSeq":
                    assert (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0, (kind, i)
                continue
            assert np.array_equal(h, eh), (kind, pk, i, q)
            if ep is not None:
                assert np.array_equal(p & L.POS_MASK, ep), (kind, pk, i)
            if kind != L.KMER:  # k-mer codes map IUPAC letters to a base (kmers.go:23-40): no flag there
                assert bool(st & L.ST_HAS_NON_ACGT) == any(c not in "ACGTacgt" for c in q) or len(q) == 0, (kind, i, q, st)
        d1 = res.digest()
        os.environ["BSK_NO_MIXED"] = "1"
        try:
            d2 = engine.run(b, engine.params(kind, **pk)).digest()
        finally:
            del os.environ["BSK_NO_MIXED"]
        assert d1 == d2, (kind, pk)


@pytest.mark.parametrize("n", [700, 40000])
def test_a_non_acgt_letter_in_every_read(engine, oracle, n):
    """Every read flagged: the mixed plan at any share where the side launch is a staged kernel (round 5; the general ASCII kernel over the
    whole batch before), and the 2-bit kernel is not launched at all -- every reference word and status byte is the side launch's."""
    rng = random.Random(n)
    seqs = []
    for i in range(n):
        q = list(rand_seq(rng, rng.choice([150, 150, rng.randint(1, 300)])))  # (below the syncmers' tile threshold: tiles carry descriptors of their own and keep the general side kernel)
        q[rng.randrange(len(q))] = rng.choice("NRYn")
        seqs.append("".join(q))
    b = engine.batch(seqs)
    step = max(1, n // 1500)
    for kind, pk, fn in ((L.MINIMIZER, dict(k=21, w=11), lambda q: oracle.minimizer(q, 21, 11, False, closed=True)),
                         (L.SYNCMER, dict(k=31, s=11), lambda q: oracle.syncmer(q, 31, 11, False, closed=True))):
        res = engine.run(b, engine.params(kind, **pk))
        assert "ASCII side launch" in res.plan()["kernel"], res.plan()
        for i in range(0, n, step):
            q = seqs[i]
            st, h, p = res.read(i)
            try:
                eh, ep, es, fl = fn(q)
            except oracle.OracleError as e:
                assert e.name == "ErrShortSeq" and (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0, (kind, i, e.name)
                continue
            assert np.array_equal(h, eh) and np.array_equal(p & L.POS_MASK, ep) and np.array_equal(p >> 31, es), (kind, i, q)
            assert st & L.ST_HAS_NON_ACGT, (kind, i, st)
        d1 = res.digest()
        res.close()
        os.environ["BSK_NO_MIXED"] = "1"  # the whole batch on the general ASCII kernel
        try:
            r2 = engine.run(b, engine.params(kind, **pk))
            assert "side launch" not in r2.plan()["kernel"], r2.plan()
            d2 = r2.digest()
            r2.close()
        finally:
            del os.environ["BSK_NO_MIXED"]
        assert d1 == d2, (kind, pk)
    b.close()


def test_protein_slab_overflow_falls_back_with_many_units(engine, oracle):
    """A low-complexity protein selects a new minimizer at every position and outgrows its slab of the register kernel: the
    call re-plans on the general kernel.  With more than 64 sequences that re-plan once kept the slab flags of the abandoned
    plan and under-sized the look-back scratch (nondeterministic results)."""
    rng = random.Random(123)
    seqs = [rand_seq(rng, rng.randint(40, 400), AA) for _ in range(400)]
    seqs[137] = "A" * 400
    seqs[300] = "AC" * 150
    b = engine.batch(seqs, L.ALPHA_PROTEIN)
    for _ in range(3):
        res = engine.run(b, engine.params(L.PROT_MINIMIZER, 9, w=5))
        for i, q in enumerate(seqs):
            st, h, p = res.read(i)
            try:
                eh, ep, fl = oracle.protein_minimizer(q, 9, 5, closed=True)
            except oracle.OracleError:
                assert (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0
                continue
            assert np.array_equal(h, eh) and np.array_equal(p & L.POS_MASK, ep), (i, len(q))


def test_long_outliers_do_not_inflate_per_read_slabs(engine, oracle):
    """Kernels with one slab per read size it by the longest read: a batch of short sequences with one long outlier must not
    reserve the outlier's slab for everybody (proteins: the batch is tiled instead; DNA: the unit-slab kernel is used)."""
    rng = random.Random(5)
    prot = [rand_seq(rng, rng.randint(30, 120), AA) for _ in range(6000)]
    prot[1234] = rand_seq(rng, 3000, AA)
    b = engine.batch(prot, L.ALPHA_PROTEIN)
    res = engine.run(b, engine.params(L.PROT_MINIMIZER, 9, w=5))
    assert b.info()["device_bytes"] < 50e6
    for i in list(range(0, 6000, 97)) + [1233, 1234, 1235]:
        st, h, p = res.read(i)
        try:
            eh, ep, _ = oracle.protein_minimizer(prot[i], 9, 5, closed=True)
        except oracle.OracleError:
            assert (st & L.ST_CODE_MASK) == L.ST_SHORT
            continue
        assert np.array_equal(h, eh) and np.array_equal(p & L.POS_MASK, ep), i
    dna = [rand_seq(rng, 150) for _ in range(4000)]
    dna[77] = rand_seq(rng, 3500)
    b = engine.batch(dna)
    res = engine.run(b, engine.params(L.MINIMIZER, 21, w=5))
    for i in (0, 76, 77, 78, 3999):
        st, h, p = res.read(i)
        eh, ep, es, _ = oracle.minimizer(dna[i], 21, 5, False, closed=True)
        assert np.array_equal(h, eh) and np.array_equal(p & L.POS_MASK, ep), i


@pytest.mark.parametrize("w", [2, 3, 4, 5, 6, 7, 8])
def test_protein_minimizer_register_kernel_whole_grid(engine, oracle, w):
    """Every window 2..8 with every k 4..16 (4..8: round 5) runs on k_prot_minimizer_fast<W,K> -- protein-fed and, for 2-bit DNA batches,
    with the translation fused into its residue fetch (all six frames): both against the oracle, and the plan is the register kernel."""
    for k in range(4, 17):
        rng = random.Random(w * 100 + k)
        prot = [rand_seq(rng, rng.choice([300, rng.randint(1, 500)]), AA) for _ in range(70)] + ["A" * 90, rand_seq(rng, 3 * k + w - 1, AA)]
        b = engine.batch(prot, L.ALPHA_PROTEIN)
        res = engine.run(b, engine.params(L.PROT_MINIMIZER, k, w=w))
        assert f"k_prot_minimizer_fast<{w},{k},false>" in res.plan()["kernel"], res.plan()
        for i, q in enumerate(prot):
            st, h, p = res.read(i)
            try:
                eh, ep, fl = oracle.protein_minimizer(q, k, w, closed=True)
            except oracle.OracleError:
                assert (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0
            else:
                assert np.array_equal(h, eh) and np.array_equal(p & L.POS_MASK, ep) and (st & 0xF0) == fl, (w, k, i)
        b.close()
        frame = (1, 2, 3, -1, -2, -3)[(w + k) % 6]
        dna = [rand_seq(rng, n) for n in (3 * k - 1, 3 * k, 3 * k + w - 2, 3 * k + w - 1, 3 * (k + w) + 1, 36 * 4 + frame % 3, 900, 901, 902)]
        dna += [rand_seq(rng, rng.randint(1, 1500)) for _ in range(60)]
        bd = engine.batch(dna)
        rd = engine.run(bd, engine.params(L.PROT_MINIMIZER, k, w=w, codon_table=1 + (k % 2) * 10, frame=frame))
        assert f"k_prot_minimizer_fast<{w},{k},true>" in rd.plan()["kernel"], rd.plan()
        for i, q in enumerate(dna):
            st, h, p = rd.read(i)
            try:
                eh, ep, fl = oracle.protein_minimizer_nt(q, k, w, 1 + (k % 2) * 10, frame)
            except oracle.OracleError:
                assert (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0, (w, k, frame, i, len(q))
            else:
                assert (st & L.ST_CODE_MASK) == 0 and np.array_equal(h, eh) and np.array_equal(p & L.POS_MASK, ep), (w, k, frame, i, len(q))
        bd.close()


@pytest.mark.parametrize("k", [11, 21, 32])
def test_every_kmer_with_its_position_w1_and_s_equal_k(engine, oracle, k):
    """minimizer w = 1 (sketch.go:218-222) and syncmer s == k (sketch.go:328-331) yield every k-mer with its index: both run on
    k_minimizer_dense<1> since round 5 (general kernels before), read by read against the oracle, short and low-complexity reads included"""
    rng = random.Random(1000 + k)
    seqs = [rand_seq(rng, rng.choice([150, 250, rng.randint(1, 600)])) for _ in range(700)] + ["A" * 200, rand_seq(rng, k - 1), rand_seq(rng, k), ""]
    b = engine.batch(seqs)
    for kind, par, ref in ((L.MINIMIZER, dict(w=1), lambda q: oracle.minimizer(q, k, 1, False, closed=True)),
                           (L.SYNCMER, dict(s=k), lambda q: oracle.syncmer(q, k, k, False, closed=True))):
        res = engine.run(b, engine.params(kind, k, **par))
        assert "k_minimizer_dense<1>" in res.plan()["kernel"], res.plan()
        for i, q in enumerate(seqs):
            st, h, p = res.read(i)
            try:
                eh, ep, es, fl = ref(q)
            except oracle.OracleError as err:
                assert (st & L.ST_CODE_MASK) != 0 and len(h) == 0, (i, len(q), err.name)
                continue
            assert np.array_equal(h, eh) and np.array_equal(p & L.POS_MASK, ep) and np.array_equal(p >> 31, es), (kind, i, len(q))
            assert (st & 0xF0) == fl, (kind, i, st, fl)
        res.close()
    b.close()
