"""The two experimental minimizer kernels must give exactly what the planned kernel gives:
BSK_SEG=1 -> k_minimizer_seg<W> (per-read slabs, a flush every few blocks, 12 waves per CU),
BSK_WPR=1 -> k_minimizer_wpr<11> (one read per wavefront, lanes = positions: the mapping of the task statement).
They are never planned without the switch; these tests keep their measured numbers honest (same tuples, same flags)."""
import os

import numpy as np
import pytest

from bio_amd import _lib as L

pytestmark = pytest.mark.gpu


def _built():
    from bio_amd import _lib
    return bool(_lib.load().bsk_build_has_experiments())


def ragged_batch(rng, n, lo, hi, frac_n=0.0):
    lens = rng.integers(lo, hi, n)
    lens[::5] = 150
    offs = np.zeros(n + 1, np.uint64)
    offs[1:] = np.cumsum(lens)
    data = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(offs[-1]))].copy()
    if frac_n:
        data[rng.integers(0, len(data), int(n * frac_n))] = ord("N")
    return data, offs


@pytest.mark.parametrize("switch,k,w", [("BSK_SEG", 21, 11), ("BSK_SEG", 15, 5), ("BSK_SEG", 31, 15), ("BSK_WPR", 21, 11), ("BSK_WPR", 5, 11),
                                        ("BSK_WPR", 54, 11), ("BSK_WPR", 33, 11)])
def test_experimental_kernels_equal_the_planned_one(engine, oracle, switch, k, w):
    if not _built():
        pytest.skip("libbiosketch.so was built without the experiments (make -C bio_amd/csrc EXPERIMENTS=1)")
    rng = np.random.default_rng(k * 100 + w)
    for n, lo, hi, fn in ((3000, 20, 400, 0.0), (777, 140, 160, 0.02), (64, 1000, 4000, 0.0), (5, 10, 30, 0.0)):
        data, offs = ragged_batch(rng, n, lo, hi, fn)
        # adversarial: runs of one letter and short periods select a position at every step / tie everywhere
        data[: int(offs[3])] = ord("A")
        seg = np.frombuffer(b"ACGTTGCA", np.uint8)
        if n >= 6:
            a, e = int(offs[3]), int(offs[6])
            data[a:e] = np.resize(seg, e - a)
        b = engine.batch_from_arrays(data, offs)
        p = engine.params(L.MINIMIZER, k, w=w)
        want = engine.run(b, p)
        os.environ[switch] = "1"
        try:
            got = engine.run(b, p)
        finally:
            del os.environ[switch]
        name = got.plan()["kernel"]
        if int(offs[1:].max() if n else 0) and max(np.diff(offs.astype(np.int64))) <= 4096:  # not tiled: the switch must have taken effect
            assert ("k_minimizer_seg" in name) if switch == "BSK_SEG" else ("k_minimizer_wpr" in name), name
        assert got.digest() == want.digest(), (switch, k, w, n)
        o1, s1, h1, p1 = want.fetch()
        o2, s2, h2, p2 = got.fetch()
        assert np.array_equal(o1, o2) and np.array_equal(s1, s2) and np.array_equal(h1, h2) and np.array_equal(p1, p2)
        # and both equal the oracle on a few reads
        for i in range(0, n, max(1, n // 7)):
            q = data[int(offs[i]):int(offs[i + 1])].tobytes()
            try:
                eh, ep, es, fl = oracle.minimizer(q, k, w, False, closed=True)
            except oracle.OracleError:
                assert (s2[i] & L.ST_CODE_MASK) == L.ST_SHORT
                continue
            lo_, hi_ = int(o2[i]), int(o2[i + 1])
            assert np.array_equal(h2[lo_:hi_], eh) and np.array_equal(p2[lo_:hi_] & L.POS_MASK, ep) and np.array_equal(p2[lo_:hi_] >> 31, es)
            assert (int(s2[i]) & 0xF0) == fl
        b.close()


@pytest.mark.parametrize("k,s,lo,hi", [(31, 11, 150, 150), (31, 11, 60, 224), (21, 11, 40, 200), (25, 12, 100, 180), (64, 44, 130, 224)])
def test_two_pass_syncmer_plan_equals_the_planned_kernels(engine, oracle, monkeypatch, k, s, lo, hi):
    """BSK_SYN_SEL=1 -> k_syncmer_sel<k - s> + k_syncmer_emit (kernels_syncmer_sel.hpp: select, then hash only what was selected --
    measured and not planned): the same tuples and flags as the one-pass kernels, read by read against the oracle."""
    if not _built():
        pytest.skip("libbiosketch.so was built without the experiments (make -C bio_amd/csrc EXPERIMENTS=1)")
    import random
    rng = random.Random(k * 100 + s + hi)
    n = 9000
    seqs = ["".join(rng.choice("ACGT") for _ in range(rng.randint(lo, hi))) for _ in range(n)]
    seqs[3] = "A" * len(seqs[3])          # a read of key ties: the exact machine's
    seqs[4] = "AC" * (len(seqs[4]) // 2)
    seqs[7] = seqs[7][: 2 * k - s - 2]    # too short: ErrShortSeq
    b = engine.batch(seqs)
    p = engine.params(L.SYNCMER, k, s=s)
    want = engine.run(b, p)
    wd = want.digest()
    monkeypatch.setenv("BSK_SYN_SEL", "1")
    got = engine.run(b, p)
    assert "k_syncmer_sel" in got.plan()["kernel"], got.plan()
    assert got.digest() == wd
    for i in list(range(0, n, 97)) + [3, 4, 7]:
        st, h, ps = got.read(i)
        st0, h0, p0 = want.read(i)
        assert st == st0 and np.array_equal(h, h0) and np.array_equal(ps, p0), i
        try:
            eh, ep, es, fl = oracle.syncmer(seqs[i], k, s, False, closed=True)
        except oracle.OracleError:
            assert (st & L.ST_CODE_MASK) == L.ST_SHORT
            continue
        assert np.array_equal(h, eh) and np.array_equal(ps & L.POS_MASK, ep) and np.array_equal(ps >> 31, es), i
    got.close()
    want.close()
    b.close()
