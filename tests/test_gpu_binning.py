"""GPU: length-binned units (k_bin_desc, KArgs::binned) -- ragged batches of short reads on the lock-step minimizer kernels.

A unit of 64 reads costs its longest read, so the reads of every chunk of 4096 are grouped by length before units are formed; the
reference words and status bytes stay at the reads' own positions.  Per-read parity against the oracle through the binned path, the
same digest as the unbinned run, a batch with non-ACGT reads (the flags travel with the permutation), the list of reads for the exact
machine (low-complexity reads inside binned units), refill of a batch object.  Contract: per-sequence independence,
sketches/sketch.go:46."""
import random

import numpy as np
import pytest

from bio_amd import _lib as L

pytestmark = pytest.mark.gpu


def rand_dna(rng, n, alpha="ACGT"):
    return "".join(rng.choice(alpha) for _ in range(n))


def check_reads(res, oracle, seqs, k, w, idx):
    for i in idx:
        s = seqs[i]
        st, h, p = res.read(i)
        try:
            eh, ep, es, fl = oracle.minimizer(s, k, w, False, closed=True)
        except oracle.OracleError as e:
            assert e.name == "ErrShortSeq"
            assert (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0, (i, s, k, w)
            continue
        assert (st & L.ST_CODE_MASK) == L.ST_OK, (i, s, st)
        assert np.array_equal(h, eh), (i, len(s), k, w)
        assert np.array_equal(p & L.POS_MASK, ep), (i, len(s), k, w)
        assert np.array_equal(p >> 31, es), (i, len(s), k, w)
        assert (st & 0xF0) == fl, (i, s, k, w, st, fl)


@pytest.mark.parametrize("k,w,lo,hi", [(21, 11, 0, 150), (21, 11, 60, 150), (15, 5, 20, 120), (31, 13, 40, 156), (21, 11, 100, 300), (21, 4, 30, 200)])
def test_binned_units_per_read_parity(engine, oracle, k, w, lo, hi):
    rng = random.Random(k * 100 + w + hi)
    n = 9000 + rng.randint(0, 700)  # chunks of 4096, the last one partial
    seqs = [rand_dna(rng, rng.randint(lo, hi)) for _ in range(n)]
    b = engine.batch(seqs)
    p = engine.params(L.MINIMIZER, k, w=w)
    res = engine.run(b, p)
    assert "length-binned" in res.plan()["kernel"], res.plan()
    check_reads(res, oracle, seqs, k, w, range(n))
    res.close()
    b.close()


@pytest.mark.parametrize("view", ["built with the batch", "built per plan"])
def test_binned_digest_equals_unbinned(engine, monkeypatch, view):
    """the length-binned view comes with the batch (bin_with_batch: classes of a base or two, no pass per plan -- bsk_batch_prepare has
    nothing to do) or, BSK_NO_BIN_EARLY=1, is built by the first plan that wants it (classes of the plan's block; the pass is what
    bsk_batch_prepare times); either way the result is the one of units in batch order"""
    if view == "built per plan":
        monkeypatch.setenv("BSK_NO_BIN_EARLY", "1")
    rng = np.random.default_rng(9)
    n = 200_000
    lens = rng.integers(60, 151, n, dtype=np.uint64)
    offs = np.zeros(n + 1, np.uint64)
    np.cumsum(lens, out=offs[1:])
    data = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(offs[-1]), dtype=np.uint8)]
    b = engine.batch_from_arrays(data, offs)
    p = engine.params(L.MINIMIZER, 21, w=11)
    if view == "built per plan":
        assert engine.prepare(b, p) > 0.0
    else:
        assert engine.prepare(b, p) == 0.0
    res = engine.run(b, p)
    assert "length-binned" in res.plan()["kernel"]
    d1 = res.digest()
    o1, s1, h1, p1 = res.fetch()
    res.close()
    ps = engine.params(L.SYNCMER, 31, s=11)  # (another plan over the same batch: the same view, or a second pass)
    rs = engine.run(b, ps)
    assert "length-binned" in rs.plan()["kernel"]
    ds = rs.digest()
    rs.close()
    monkeypatch.setenv("BSK_NO_BIN", "1")
    assert engine.prepare(b, p) == 0.0
    res = engine.run(b, p)
    assert "length-binned" not in res.plan()["kernel"]
    d2 = res.digest()
    o2, s2, h2, p2 = res.fetch()
    res.close()
    rs = engine.run(b, ps)
    assert "length-binned" not in rs.plan()["kernel"] and rs.digest() == ds
    rs.close()
    b.close()
    assert d1 == d2
    assert np.array_equal(o1, o2) and np.array_equal(s1, s2) and np.array_equal(h1, h2) and np.array_equal(p1, p2)


def test_binned_units_with_flagged_and_low_complexity_reads(engine, oracle):
    """Reads with an N (mixed batch: 2-bit kernel over everything + ASCII side launch, the input flags follow the permutation) and
    poly-A tails (key ties: the list of reads for the exact machine names SLOTS of the binned order)."""
    rng = random.Random(77)
    seqs = []
    for i in range(2600):
        s = rand_dna(rng, rng.randint(50, 150))
        if i % 97 == 0 and len(s) > 30:
            s = s[:17] + "N" + s[18:]
        if i % 13 == 0:
            s = s[: len(s) // 2] + "A" * (len(s) - len(s) // 2)
        seqs.append(s)
    b = engine.batch(seqs)
    p = engine.params(L.MINIMIZER, 21, w=11)
    res = engine.run(b, p)
    assert "length-binned" in res.plan()["kernel"], res.plan()
    check_reads(res, oracle, seqs, 21, 11, range(len(seqs)))
    for i in range(0, len(seqs), 97):
        if "N" in seqs[i]:
            assert res.read(i)[0] & L.ST_HAS_NON_ACGT
    res.close()
    b.close()


def test_binned_view_is_rebuilt_per_window_and_after_refill(engine, oracle):
    rng = random.Random(5)
    seqs = [rand_dna(rng, rng.randint(40, 150)) for _ in range(2048)]
    b = engine.batch(seqs)
    for k, w in ((21, 11), (21, 5), (21, 11)):  # the class width follows w: the view is rebuilt
        res = engine.run(b, engine.params(L.MINIMIZER, k, w=w))
        assert "length-binned" in res.plan()["kernel"]
        check_reads(res, oracle, seqs, k, w, range(0, len(seqs), 7))
        res.close()
    b.close()


@pytest.mark.parametrize("k,s,lo,hi", [(31, 11, 40, 150), (21, 11, 30, 200), (21, 11, 30, 224), (15, 9, 20, 400)])
def test_binned_units_syncmers(engine, oracle, k, s, lo, hi):
    """The syncmer kernels (k_syncmer_pk / k_syncmer_pkl + their list pass) over length-binned units -- and, since round 5, ragged batches
    whose longest read is beyond the long packed plan's columns (209 bases at k = 21 s = 11; dense selections at k = 15 s = 9) over tiles"""
    rng = random.Random(k * 100 + s)
    n = 9000 + rng.randint(0, 500)
    seqs = [rand_dna(rng, rng.randint(lo, hi)) for _ in range(n)]
    b = engine.batch(seqs)
    res = engine.run(b, engine.params(L.SYNCMER, k, s=s))
    assert ("length-binned" if hi <= 200 else "over tiles") in res.plan()["kernel"], res.plan()
    for i in range(0, n, 3):
        st, h, p = res.read(i)
        try:
            eh, ep, es, fl = oracle.syncmer(seqs[i], k, s, False, closed=True)
        except oracle.OracleError as e:
            assert e.name == "ErrShortSeq" and (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0, (i, len(seqs[i]))
            continue
        assert np.array_equal(h, eh) and np.array_equal(p & L.POS_MASK, ep) and np.array_equal(p >> 31, es), (i, len(seqs[i]), k, s)
        assert (st & 0xF0) == fl, (i, st, fl)
    res.close()
    b.close()
