"""GPU: class plans (run_classed, classes.hip) -- one plan per LENGTH CLASS of a batch instead of one plan keyed on the longest read.

The reference sketches one sequence at a time: a 5-kb contig costs its own 5 kb and nothing else (sketches/sketch.go:46, :85-94).  A batch
of 150-base reads with a few longer ones must therefore (a) give every read exactly the tuples of its own iterator -- per-read parity
against the oracle, through >= 2 kernels -- and (b) keep the digest of the one-plan run (BSK_NO_CLASS)."""
import random
import zlib

import numpy as np
import pytest

from bio_amd import _lib as L

pytestmark = pytest.mark.gpu


def rand_dna(rng, n, alpha="ACGT"):
    return "".join(rng.choice(alpha) for _ in range(n))


def check_min(res, oracle, seqs, k, w, idx):
    for i in idx:
        s = seqs[i]
        st, h, p = res.read(i)
        if "N" in s:
            assert st & L.ST_HAS_NON_ACGT, (i, st)
        try:
            eh, ep, es, fl = oracle.minimizer(s, k, w, False, closed=True)
        except oracle.OracleError as e:
            assert e.name == "ErrShortSeq"
            assert (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0, (i, len(s), k, w)
            continue
        assert (st & L.ST_CODE_MASK) == L.ST_OK, (i, len(s), st)
        assert np.array_equal(h, eh), (i, len(s), k, w)
        assert np.array_equal(p & L.POS_MASK, ep), (i, len(s), k, w)
        assert np.array_equal(p >> 31, es), (i, len(s), k, w)


def check_syn(res, oracle, seqs, k, s_, idx):
    for i in idx:
        st, h, p = res.read(i)
        try:
            eh, ep, es, fl = oracle.syncmer(seqs[i], k, s_, False, closed=True)
        except oracle.OracleError as e:
            assert e.name == "ErrShortSeq"
            assert (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0
            continue
        assert np.array_equal(h, eh), (i, len(seqs[i]))
        assert np.array_equal(p & L.POS_MASK, ep) and np.array_equal(p >> 31, es), (i, len(seqs[i]))


def outlier_batch(rng, n, base_len, outliers, with_n=False, ragged=False):
    """n reads of base_len bases (ragged: 60..base_len), plus `outliers` = [(count, lo, hi)] reads of other lengths at random places"""
    seqs = [rand_dna(rng, rng.randint(60, base_len) if ragged else base_len) for _ in range(n)]
    for cnt, lo, hi in outliers:
        for _ in range(cnt):
            seqs[rng.randrange(n)] = rand_dna(rng, rng.randint(lo, hi))
    if with_n:
        for _ in range(n // 40):
            i = rng.randrange(n)
            s = seqs[i]
            if len(s) > 3:
                j = rng.randrange(len(s))
                seqs[i] = s[:j] + "N" + s[j + 1:]
    if with_n:  # ... and in every second outlier (the bulk's ASCII side launch owns the flagged reads of EVERY class: its slabs follow the batch's longest read)
        for i, s in enumerate(seqs):
            if len(s) > base_len and i % 2 == 0:
                j = rng.randrange(len(s))
                seqs[i] = s[:j] + "N" + s[j + 1:]
    # low-complexity reads in every class (the exact machine's list)
    seqs[5] = "A" * len(seqs[5])
    return seqs


CASES = [
    ("150 + a few 400", 20000, 150, [(7, 400, 400)], False, False),
    ("150 + a few 5000", 20000, 150, [(3, 5000, 5000)], False, False),
    ("150 + 1 % of 250", 20000, 150, [(200, 250, 250)], False, False),
    ("150 + 400 + 5000 + 12000, reads with N", 20000, 150, [(9, 380, 420), (3, 4000, 6000), (1, 12000, 12000)], True, False),
    ("ragged 60..150 + 300 + 3000", 20000, 150, [(50, 280, 330), (4, 3000, 3000)], False, True),
    ("250 + a few 150 and 2000", 18000, 250, [(100, 150, 150), (5, 2000, 2000)], False, False),
]


@pytest.mark.parametrize("cut", ["host list", "device pass"])
@pytest.mark.parametrize("name,n,base,outl,with_n,ragged", CASES, ids=[c[0] for c in CASES])
def test_class_plan_per_read_parity_minimizer(engine, oracle, monkeypatch, name, n, base, outl, with_n, ragged, cut):
    """cut: the other classes' lists picked on the host from the batch's list of odd sequences + the bulk's kernel masking by length
    (no device pass), or k_class_cut + a view of the batch (BSK_CLASS_VIEW; also what ragged bulks take by themselves)"""
    monkeypatch.setenv("BSK_CLASS_FORCE", "1")  # (a batch this small is not worth two launches: the cost model would keep one plan)
    if cut == "device pass":
        monkeypatch.setenv("BSK_CLASS_VIEW", "1")
    rng = random.Random(zlib.crc32(name.encode()))
    seqs = outlier_batch(rng, n, base, outl, with_n, ragged)
    b = engine.batch(seqs)
    p = engine.params(L.MINIMIZER, 21, w=11)
    res = engine.run(b, p)
    plan = res.plan()["kernel"]
    assert plan.count("k_minimizer") >= 2 and " reads of " in plan, plan  # >= 2 kernels over one result
    idx = list(range(0, n, 37)) + [i for i, s in enumerate(seqs) if len(s) != base or "N" in s] + [5]
    check_min(res, oracle, seqs, 21, 11, sorted(set(idx)))
    d = res.digest()
    # the consumers of a result see one result: dense copy, narrow fetch and sets agree with the wide fetch
    off, st, h, pos = res.fetch()
    if max(len(s) for s in seqs) < 32768:
        o32, st2, h2, p16 = res.fetch_narrow()
        assert np.array_equal(o32.astype(np.uint64), off) and np.array_equal(st2[:n], st[:n]) and np.array_equal(h2[: int(off[-1])], h[: int(off[-1])])
        assert np.array_equal(p16[: int(off[-1])] & 0x7FFF, pos[: int(off[-1])] & L.POS_MASK)
    s_off, s_val = res.sets()
    for r in (0, 5, n // 2, n - 1):
        assert np.array_equal(np.unique(h[int(off[r]):int(off[r + 1])]), s_val[int(s_off[r]):int(s_off[r + 1])])
    # the timed re-run (the bench's pattern) repeats the class plan
    res2, ms = engine.run_timed(b, p, 0, 2, reuse=res)
    assert res2.digest() == d and res2.plan()["kernel"] == plan
    res.close()
    # one plan per batch gives the same digest
    monkeypatch.delenv("BSK_CLASS_FORCE")
    monkeypatch.setenv("BSK_NO_CLASS", "1")
    one = engine.run(b, p)
    assert " reads of " not in one.plan()["kernel"]
    assert one.digest() == d
    one.close()
    b.close()


@pytest.mark.parametrize("outl", [[(6, 300, 300)], [(4, 2000, 2500)], [(150, 230, 260), (2, 700, 700)]])
def test_class_plan_per_read_parity_syncmer(engine, oracle, monkeypatch, outl):
    monkeypatch.setenv("BSK_CLASS_FORCE", "1")
    rng = random.Random(len(outl) * 7 + outl[0][1])
    n = 18000
    seqs = outlier_batch(rng, n, 150, outl)
    b = engine.batch(seqs)
    p = engine.params(L.SYNCMER, 31, s=11)
    res = engine.run(b, p)
    plan = res.plan()["kernel"]
    assert plan.count("k_syncmer") >= 2 and " reads of " in plan, plan
    idx = sorted(set(list(range(0, n, 41)) + [i for i, s in enumerate(seqs) if len(s) != 150]))
    check_syn(res, oracle, seqs, 31, 11, idx)
    d = res.digest()
    res.close()
    monkeypatch.delenv("BSK_CLASS_FORCE")
    monkeypatch.setenv("BSK_NO_CLASS", "1")
    one = engine.run(b, p)
    assert one.digest() == d
    one.close()
    b.close()


def test_class_plan_masked_bulk_through_length_binned_units(engine, oracle, monkeypatch):
    """scripts/fuzz_class.py seed 27: a bulk that is itself ragged (here reads of 144..159 bases: one bucket of the batch's length histogram,
    so the other classes come from the host's list and the bulk's kernel masks them by length) runs on length-binned units -- the binning pass
    has to mask too, or the high bits of a 4 998-base length land in the place field of the binned descriptor and a read of the bulk
    loses its reference word"""
    monkeypatch.setenv("BSK_CLASS_FORCE", "1")
    rng = random.Random(27)
    n = 17000
    seqs = [rand_dna(rng, rng.randint(144, 159)) for _ in range(n)]
    for cnt, ln in ((40, 300), (3, 700), (1, 4998), (2, 9000)):
        for _ in range(cnt):
            seqs[rng.randrange(n)] = rand_dna(rng, ln)
    seqs[5] = "A" * len(seqs[5])
    b = engine.batch(seqs)
    p = engine.params(L.MINIMIZER, 15, w=5)
    res = engine.run(b, p)
    plan = res.plan()["kernel"]
    assert "length-binned" in plan.split(" + ")[0] and " reads of " in plan, plan
    off, st, h, pos = res.fetch()
    assert not np.any((st[:n] & L.ST_CODE_MASK) == L.ST_SHORT)
    check_min(res, oracle, seqs, 15, 5, sorted(set(list(range(0, n, 53)) + [i for i, s in enumerate(seqs) if len(s) > 159] + [5])))
    d = res.digest()
    res.close()
    monkeypatch.delenv("BSK_CLASS_FORCE")
    monkeypatch.setenv("BSK_NO_CLASS", "1")
    one = engine.run(b, p)
    assert one.digest() == d
    one.close()
    b.close()


def test_class_plan_long_odd_list_is_split_on_the_device(engine, oracle, monkeypatch):
    """65 536 odd sequences or more: the classes' lists come from the batch's list ON THE DEVICE (k_odd_split: one atomic per class and
    wavefront, descriptors gathered on the way) instead of a host loop + copy (0.9 ms per 10^6 odd reads on every bsk_sketch)"""
    monkeypatch.setenv("BSK_CLASS_FORCE", "1")
    rng = np.random.default_rng(65536)
    n = 1_500_000
    lens = np.full(n, 150, np.uint64)
    lens[rng.integers(0, n, 72_000)] = 250
    lens[rng.integers(0, n, 300)] = 400
    lens[rng.integers(0, n, 3)] = 5000
    assert 65536 <= int((lens != 150).sum()) <= n // 20
    offs = np.zeros(n + 1, np.uint64)
    np.cumsum(lens, out=offs[1:])
    data = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(offs[-1]), dtype=np.uint8)]
    b = engine.batch_from_arrays(data, offs)
    p = engine.params(L.MINIMIZER, 21, w=11)
    res = engine.run(b, p)
    plan = res.plan()["kernel"]
    assert plan.count(" reads of ") >= 3, plan
    odd = np.nonzero(lens != 150)[0]
    idx = sorted(set(int(i) for i in np.concatenate([odd[:: max(1, len(odd) // 300)], np.nonzero(lens > 250)[0][:40], np.arange(0, n, n // 200)])))
    seqs = {i: data[int(offs[i]):int(offs[i + 1])].tobytes().decode() for i in idx}
    for i in idx:
        st, h, pos = res.read(i)
        eh, ep, es, fl = oracle.minimizer(seqs[i], 21, 11, False, closed=True)
        assert np.array_equal(h, eh) and np.array_equal(pos & L.POS_MASK, ep) and np.array_equal(pos >> 31, es), (i, int(lens[i]))
    d = res.digest()
    res2, ms = engine.run_timed(b, p, 0, 2, reuse=res)  # (the lists are split again on every call: any order inside a class gives the same result)
    assert res2.digest() == d
    res.close()
    monkeypatch.delenv("BSK_CLASS_FORCE")
    monkeypatch.setenv("BSK_NO_CLASS", "1")
    one = engine.run(b, p)
    assert one.digest() == d
    one.close()
    b.close()


def test_class_plan_is_not_taken_when_it_cannot_pay(engine):
    """uniform batches, small batches, a bulk that is itself tile work: one plan as before"""
    rng = random.Random(3)
    p = engine.params(L.MINIMIZER, 21, w=11)
    for seqs in ([rand_dna(rng, 150) for _ in range(20000)],                                   # one length
                 [rand_dna(rng, 150) for _ in range(2000)] + [rand_dna(rng, 900)],            # small batch
                 [rand_dna(rng, rng.randint(60, 150)) for _ in range(20000)],                  # ragged, one kernel
                 [rand_dna(rng, 6000) for _ in range(1000)] + [rand_dna(rng, 150) for _ in range(17000)]):  # most bases are tile work
        b = engine.batch(seqs)
        res = engine.run(b, p)
        assert " reads of " not in res.plan()["kernel"], res.plan()
        res.close()
        b.close()


def test_class_plan_by_the_cost_model(engine, monkeypatch):
    """a batch large enough for the cost model itself to cut it: 3 10^6 reads of 150 bases + 0.01 % of 400 and of 5 000 bases"""
    rng = np.random.default_rng(11)
    n = 3_000_000
    lens = np.full(n, 150, np.uint64)
    lens[rng.integers(0, n, 300)] = 400
    lens[rng.integers(0, n, 300)] = 5000
    offs = np.zeros(n + 1, np.uint64)
    np.cumsum(lens, out=offs[1:])
    data = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(offs[-1]), dtype=np.uint8)].copy()
    p = engine.params(L.MINIMIZER, 21, w=11)
    b = engine.batch_from_arrays(data, offs)
    res = engine.run(b, p)
    plan = res.plan()["kernel"]
    assert plan.startswith("k_minimizer_pk<11") and plan.count(" reads of ") == 2, plan
    d = res.digest()
    res.close()
    monkeypatch.setenv("BSK_NO_CLASS", "1")
    one = engine.run(b, p)
    assert "over tiles" in one.plan()["kernel"] and one.digest() == d
    one.close()
    b.close()


def test_class_plan_through_refill_and_pipeline(engine, monkeypatch):
    """chunks of a stream: every chunk cut on its own, parts re-used from chunk to chunk; the sink delivers what one batch gives"""
    from bio_amd import sketches as S
    monkeypatch.setenv("BSK_CLASS_FORCE", "1")
    rng = np.random.default_rng(5)
    n = 60_000
    lens = np.full(n, 150, np.uint64)
    lens[rng.integers(0, n, 12)] = 420
    lens[rng.integers(0, n, 3)] = 5200
    offs = np.zeros(n + 1, np.uint64)
    np.cumsum(lens, out=offs[1:])
    data = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(offs[-1]), dtype=np.uint8)].copy()
    p = engine.params(L.MINIMIZER, 21, w=11)
    whole = engine.run(engine.batch_from_arrays(data, offs), p)
    assert " reads of " in whole.plan()["kernel"]
    want = whole.digest()
    st = S.Engine.pipeline_memory(data, offs, p, n_streams=2, chunk_records=20000, fetch=True)
    assert st["tuples"] == want["n_tuples"] and st["checksum"] == want["checksum"]


@pytest.mark.parametrize("bits,what", [(2, "a part overflows while the plan is sized"), (4, "a part overflows in a timed re-run"),
                                        (1, "the bulk's region overflows in a timed re-run")])
def test_overflow_on_the_launch_the_caller_sees_is_sized_again(engine, oracle, monkeypatch, bits, what):
    """a class plan's parts are sized by launches of their own and run AGAIN inside the parent's launch; region and
    list use vary from launch to launch, so an overflow there must be seen (k_fold_flags -> the parent's read-back) and answered by sizing
    again -- not adopted with truncated tuples.  BSK_TEST_OVERFLOW pretends the flag once per call; the result must equal the one-plan run's
    read by read, and a timed re-run must succeed (bsk_sketch_timed re-sizes instead of failing)."""
    monkeypatch.setenv("BSK_CLASS_FORCE", "1")
    rng = random.Random(600 + bits)
    n = 20000
    seqs = outlier_batch(rng, n, 150, [(9, 400, 400), (150, 250, 250)])
    b = engine.batch(seqs)
    p = engine.params(L.MINIMIZER, 21, w=11)
    want = engine.run(b, p)
    d = want.digest()
    plan = want.plan()["kernel"]
    assert " reads of " in plan
    monkeypatch.setenv("BSK_TEST_OVERFLOW", str(bits))
    res = engine.run(b, p)
    assert res.digest() == d and res.plan()["kernel"] == plan, what
    res2, ms = engine.run_timed(b, p, 1, 3, reuse=res)
    assert res2.digest() == d and len(ms) == 3 and all(m > 0 for m in ms), what
    idx = sorted(set(list(range(0, n, 97)) + [i for i, s in enumerate(seqs) if len(s) != 150]))
    check_min(res2, oracle, seqs, 21, 11, idx)
    # and without a class plan: the timed re-run of a slab kernel sizes again once
    if bits == 1:
        monkeypatch.delenv("BSK_CLASS_FORCE")
        monkeypatch.setenv("BSK_NO_CLASS", "1")
        u = engine.synth(L.ALPHA_DNA, 30000, 150, 77)
        r0 = engine.run(u, p)
        d0 = r0.digest()
        r1, ms = engine.run_timed(u, p, 0, 2, reuse=r0)
        assert r1.digest() == d0 and all(m > 0 for m in ms)
        u.close()
    res.close()
    want.close()
    b.close()
