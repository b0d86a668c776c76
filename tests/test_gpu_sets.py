"""Sketch sets on the device (bsk_result_sets): sorted distinct hash values per sequence / per batch, FracMinHash filter."""
import os
import random

import numpy as np
import pytest

from bio_amd import _lib as L

pytestmark = pytest.mark.gpu


def rand_seq(rng, n, alpha="ACGT"):
    return "".join(rng.choice(alpha) for _ in range(n))


def expect_sets(res, n, scale, whole):
    maxhash = (2**64 - 1) // scale if scale > 1 else 2**64 - 1
    per = []
    for i in range(n):
        _, h, _ = res.read(i)
        per.append(np.unique(h[h <= np.uint64(maxhash)]))
    if whole:
        return [np.unique(np.concatenate(per)) if per else np.zeros(0, np.uint64)]
    return per


@pytest.mark.parametrize("kind,pk", [(L.MINIMIZER, dict(k=21, w=11)), (L.SYNCMER, dict(k=31, s=11)), (L.NTHASH, dict(k=21)),
                                     (L.MINIMIZER, dict(k=5, w=3)), (L.KMER, dict(k=4))])
@pytest.mark.parametrize("scale", [1, 7])
def test_sets_equal_numpy_unique(engine, kind, pk, scale):
    rng = random.Random(len(pk) * 10 + scale)
    seqs = [rand_seq(rng, rng.choice([150, 150, rng.randint(1, 400)])) for _ in range(500)]
    seqs += ["", "A" * 300, "AC" * 200, "ACGTTGCAACGT" * 30, rand_seq(rng, 200, "ACGTN")]  # heavy duplicates, empty sets
    b = engine.batch(seqs)
    res = engine.run(b, engine.params(kind, **pk))
    for whole in (False, True):
        offs, vals = res.sets(whole_batch=whole, scale=scale)
        want = expect_sets(res, len(seqs), scale, whole)
        assert len(offs) == len(want) + 1 and int(offs[-1]) == len(vals)
        for i, w in enumerate(want):
            got = vals[int(offs[i]):int(offs[i + 1])]
            assert np.array_equal(got, w), (kind, pk, scale, whole, i, len(got), len(w))


def oracle_values(oracle, kind, pk, q):
    """Next*() values of the reference iterator / sketch for one sequence, from the CPU oracle ([] where the constructor fails)."""
    try:
        if kind == L.MINIMIZER:
            return oracle.minimizer(q, pk["k"], pk["w"], False, closed=True)[0]
        if kind == L.SYNCMER:
            return oracle.syncmer(q, pk["k"], pk["s"], False, closed=True)[0]
        if kind == L.NTHASH:
            return oracle.nthash(q, pk["k"], True)[0]
        return oracle.kmer_codes(q, pk["k"], True, False)
    except oracle.OracleError:
        return np.zeros(0, np.uint64)


@pytest.mark.parametrize("kind,pk", [(L.MINIMIZER, dict(k=21, w=11)), (L.SYNCMER, dict(k=31, s=11)), (L.NTHASH, dict(k=21)),
                                     (L.MINIMIZER, dict(k=7, w=4)), (L.KMER, dict(k=6))])
@pytest.mark.parametrize("scale", [1, 10])
def test_sets_equal_the_oracles_sorted_distinct_values(engine, oracle, kind, pk, scale):
    """What kmcp / unikmer keep of a sketch: collect every Next*() value of the reference iterator, drop those above
    MaxUint64/scale (the FracMinHash rule of iterator.go:181-185), sort, de-duplicate -- computed here from the CPU ORACLE's values,
    not from the engine's own tuples: per sequence (both the one-group-per-sequence bitonic kernel for small counts and the
    segmented radix sort) and for the whole batch."""
    rng = random.Random(kind * 100 + scale)
    seqs = [rand_seq(rng, rng.choice([150, 150, rng.randint(1, 400)])) for _ in range(300)]
    seqs += ["", "A" * 300, "AC" * 200, "ACGTTGCAACGT" * 30, rand_seq(rng, 3000), rand_seq(rng, 9000)]  # duplicates; counts above 64
    maxhash = np.uint64((2**64 - 1) // scale)
    per = []
    for q in seqs:
        v = np.asarray(oracle_values(oracle, kind, pk, q), np.uint64)
        per.append(np.unique(v[v <= maxhash]))
    b = engine.batch(seqs)
    res = engine.run(b, engine.params(kind, **pk))
    offs, vals = res.sets(whole_batch=False, scale=scale)
    assert len(offs) == len(seqs) + 1 and int(offs[-1]) == len(vals)
    for i, w in enumerate(per):
        assert np.array_equal(vals[int(offs[i]):int(offs[i + 1])], w), (kind, pk, scale, i, len(w))
    offs, vals = res.sets(whole_batch=True, scale=scale)
    assert list(offs) == [0, len(vals)] and np.array_equal(vals, np.unique(np.concatenate(per)))
    # short reads only: every count <= 64, the small-set kernel alone
    b2 = engine.batch(seqs[:300])
    res2 = engine.run(b2, engine.params(kind, **pk))
    offs, vals = res2.sets(whole_batch=False, scale=scale)
    for i, w in enumerate(per[:300]):
        assert np.array_equal(vals[int(offs[i]):int(offs[i + 1])], w), ("small", kind, pk, scale, i)


def test_sets_of_tiled_and_mixed_results(engine):
    os.environ["BSK_TILE_MIN"] = "64"
    try:
        rng = random.Random(5)
        seqs = [rand_seq(rng, 3000), rand_seq(rng, 40), rand_seq(rng, 1500)[:700] + "N" + rand_seq(rng, 800), rand_seq(rng, 5000)]
        b = engine.batch(seqs)
        for kind, pk in ((L.MINIMIZER, dict(k=15, w=8)), (L.NTHASH, dict(k=11))):
            res = engine.run(b, engine.params(kind, **pk))  # wide (tiled) result
            offs, vals = res.sets(scale=3)
            for i, w in enumerate(expect_sets(res, len(seqs), 3, False)):
                assert np.array_equal(vals[int(offs[i]):int(offs[i + 1])], w), (kind, i)
    finally:
        del os.environ["BSK_TILE_MIN"]


def test_sets_empty_result(engine):
    b = engine.batch(["ACGT", "", "AC"])
    res = engine.run(b, engine.params(L.MINIMIZER, 21, w=11))  # everything too short
    offs, vals = res.sets()
    assert list(offs) == [0, 0, 0, 0] and len(vals) == 0
    offs, vals = res.sets(whole_batch=True)
    assert list(offs) == [0, 0] and len(vals) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["min", "nt", "long"])
def test_compact_device_copy_equals_fetch(engine, kind):
    """bsk_result_compact leaves on the device exactly what bsk_result_fetch brings to the host (slab gaps squeezed out; wide results too)."""
    import ctypes as C
    rng = random.Random(31)
    if kind == "long":
        seqs = ["".join(rng.choice("ACGT") for _ in range(n)) for n in (70000, 150, 9000, 0, 33000)]
        p = engine.params(L.MINIMIZER, 21, w=11)
    else:
        seqs = ["".join(rng.choice("ACGT") for _ in range(rng.choice([150, 150, 90, 20, 251]))) for _ in range(700)]
        p = engine.params(L.MINIMIZER, 21, w=11) if kind == "min" else engine.params(L.NTHASH, 21)
    res = engine.run(engine.batch(seqs), p)
    offs, _, h, pos = res.fetch()
    po, ph, pp, nt = res.compact()
    assert nt == len(h) == int(offs[-1])
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def d2h(ptr, n, dt):
        a = np.empty(n, dt)
        if n:
            assert hip.hipMemcpy(a.ctypes.data, ptr, a.nbytes, 2) == 0
        return a

    assert np.array_equal(d2h(po, len(seqs) + 1, np.uint64), offs)
    assert np.array_equal(d2h(ph, nt, np.uint64), h)
    if pos is None:
        assert pp is None
    else:
        assert np.array_equal(d2h(pp, nt, np.uint32), pos)


@pytest.mark.gpu
@pytest.mark.parametrize("lens,plan", [((200, 250, 290), "k_minimizer_ring"), ((300, 420), "k_minimizer_pkd"), ((150,), "k_minimizer_pk"), ((60, 100, 150), "k_minimizer_pk"), ((40, 900, 2500), "minimizer")])
def test_group_gather_layouts_and_mixed_groups(engine, lens, plan):
    """bsk_result_compact / bsk_result_fetch_narrow gather by GROUPS of 64 sequences (k_gather_groups: offsets searched with ds_bpermute,
    unit rows through an LDS image): against bsk_result_fetch's own wavefront-per-sequence gather on slab, unit-row and per-read-slab
    results, with listed reads (poly-A: stride 1 inside a unit of rows), sequences without tuples, a last group that is not full,
    ranges that do not start on a group boundary -- and against the per-sequence kernels they replaced (BSK_NO_GROUP_GATHER)."""
    import ctypes as C
    rng = random.Random(len(lens) * 1000 + lens[0])
    n = 64 * 37 + 29
    seqs = ["".join(rng.choice("ACGT") for _ in range(rng.choice(lens))) for _ in range(n)]
    for i in range(5, n, 97):
        seqs[i] = "A" * len(seqs[i])          # listed: the exact machine writes it elsewhere, stride 1
    for i in range(11, n, 131):
        seqs[i] = seqs[i][:17]                # too short: no tuples
    seqs[64], seqs[65], seqs[127] = "", "ACGT", ""
    b = engine.batch(seqs)
    res = engine.run(b, engine.params(L.MINIMIZER, 21, w=11))
    assert plan in res.plan()["kernel"], res.plan()
    offs, st, h, pos = res.fetch()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def d2h(ptr, m, dt):
        a = np.empty(m, dt)
        if m:
            assert hip.hipMemcpy(a.ctypes.data, ptr, a.nbytes, 2) == 0
        return a

    def check():
        po, ph, pp, nt = res.compact()
        assert nt == len(h)
        assert np.array_equal(d2h(po, n + 1, np.uint64), offs) and np.array_equal(d2h(ph, nt, np.uint64), h) and np.array_equal(d2h(pp, nt, np.uint32), pos)
        for first, count in ((0, None), (1, 63), (63, 130), (64 * 20 + 7, 64 * 9 + 1), (n - 1, 1), (n - 30, 30), (9, 0)):
            o, s1, hh, p = res.fetch(first, count)
            o2, s2, h2, p2 = res.fetch_narrow(first, count)
            assert np.array_equal(o, o2.astype(np.uint64)) and np.array_equal(s1, s2) and np.array_equal(hh, h2), (first, count)
            assert np.array_equal(p & L.POS_MASK, (p2 & 0x7FFF).astype(np.uint32)) and np.array_equal(p >> 31, (p2 >> 15).astype(np.uint32)), (first, count)

    check()
    os.environ["BSK_NO_GROUP_GATHER"] = "1"
    try:
        engine.reload_options()
        check()
    finally:
        del os.environ["BSK_NO_GROUP_GATHER"]
        engine.reload_options()
    res.close()
    b.close()
