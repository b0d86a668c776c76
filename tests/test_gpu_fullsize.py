"""GPU: BASELINE-size runs checked through size-independent properties + sampled bit-exact parity."""
import os

import numpy as np
import pytest

from bio_amd import _lib as L

pytestmark = pytest.mark.gpu
SEED = 0x5EED0003


def _sample_parity(oracle, batch, res, first, count, k, w):
    data, offs = batch.fetch_ascii(first, count)
    o, st, h, p = res.fetch(first, count)
    for r in range(count):
        s = data[int(offs[r]):int(offs[r + 1])].tobytes()
        eh, ep, es, fl = oracle.minimizer(s, k, w, closed=True)
        a, e = int(o[r]), int(o[r + 1])
        assert np.array_equal(h[a:e], eh) and np.array_equal(p[a:e] & L.POS_MASK, ep) and np.array_equal(p[a:e] >> 31, es), (first, r)
        assert (int(st[r]) & 0xF0) == fl


def test_config3_100M_reads_minimizer_k21_w11(engine, oracle):
    """BASELINE configs[2] at full size: 100M x 150 bp, k=21 w=11."""
    n = 100_000_000
    b = engine.synth(L.ALPHA_DNA, n, 150, SEED)
    res = engine.run(b, engine.params(L.MINIMIZER, 21, w=11))
    d1 = res.digest()
    assert d1["short"] == 0 and d1["illegal"] == 0 and d1["has_non_acgt"] == 0
    assert d1["n_tuples"] == res.info()["n_tuples"]
    assert 22.0 < d1["n_tuples"] / n < 22.25  # density 2/(w+1) of 120 windows (+ the first window), measured 22.11
    # idempotence / determinism: a second run into the same buffers gives the identical digest
    res = engine.run(b, engine.params(L.MINIMIZER, 21, w=11), reuse=res)
    assert res.digest() == d1
    # sampled bit-exact parity at the start, in the middle and at the very end of the batch
    for first in (0, 49_999_000, n - 1500):
        _sample_parity(oracle, b, res, first, 1500, 21, 11)
    # counter-based generator: the first 200k reads of this batch are the 200k-read batch with the same seed
    small = engine.synth(L.ALPHA_DNA, 200_000, 150, SEED)
    rs = engine.run(small, engine.params(L.MINIMIZER, 21, w=11))
    data, offs = small.fetch_ascii(0, 200_000)
    nt, ck = oracle.batch_run(4, data, offs, 21, 11, threads=os.cpu_count() or 1)
    ds = rs.digest()
    assert (ds["n_tuples"], ds["checksum"]) == (nt, ck)
    o1, _, h1, p1 = res.fetch(0, 200_000)
    o2, _, h2, p2 = rs.fetch(0, 200_000)
    assert np.array_equal(o1, o2) and np.array_equal(h1, h2) and np.array_equal(p1, p2)


def test_config2_10M_reads_nthash_stream(engine, oracle):
    """BASELINE configs[1] at full size: 10M x 150 bp, canonical ntHash k=21."""
    n = 10_000_000
    b = engine.synth(L.ALPHA_DNA, n, 150, SEED)
    res = engine.run(b, engine.params(L.NTHASH, 21))
    d = res.digest()
    assert d["n_tuples"] == n * 130 and d["short"] == 0
    for first in (0, n - 2000):
        data, offs = b.fetch_ascii(first, 2000)
        o, st, h, _ = res.fetch(first, 2000)
        for r in range(0, 2000, 7):
            eh, _ = oracle.nthash(data[int(offs[r]):int(offs[r + 1])].tobytes(), 21)
            assert np.array_equal(h[int(o[r]):int(o[r + 1])], eh)
    # linearity across kernels: every minimizer tuple is the stream value at its position
    rm = engine.run(b, engine.params(L.MINIMIZER, 21, w=11))
    o, _, h, _ = res.fetch(5_000_000, 3000)
    om, _, hm, pm = rm.fetch(5_000_000, 3000)
    for r in range(3000):
        pos = pm[int(om[r]):int(om[r + 1])] & L.POS_MASK
        assert np.all(np.diff(pos.astype(np.int64)) > 0)  # positions strictly increasing
        assert np.array_equal(hm[int(om[r]):int(om[r + 1])], h[int(o[r]) + pos])
    # fast (W-specialised, slab output) and generic (run-time w, look-back CSR) kernels agree on the whole batch
    dm = rm.digest()
    os.environ["BSK_FORCE_GENERIC"] = "1"
    try:
        dg = engine.run(b, engine.params(L.MINIMIZER, 21, w=11)).digest()
        dn = engine.run(b, engine.params(L.NTHASH, 21)).digest()
    finally:
        del os.environ["BSK_FORCE_GENERIC"]
    assert dg == dm and dn == d


def test_ragged_batch_matches_uniform(engine):
    """the per-lane bound path (ragged lengths) and the wave-uniform path give the same per-read tuples"""
    b = engine.synth(L.ALPHA_DNA, 100_000, 150, SEED)
    data, offs = b.fetch_ascii(0, 100_000)
    ru = engine.run(b, engine.params(L.MINIMIZER, 21, w=11))
    seqs = [data[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(0, 100_000, 50)]
    seqs2 = []
    for i, s in enumerate(seqs):  # interleave shorter reads so that no wave is uniform
        seqs2.append(s)
        seqs2.append(s[: 40 + i % 100])
    rb = engine.run(engine.batch(seqs2), engine.params(L.MINIMIZER, 21, w=11))
    ou, _, hu, pu = ru.fetch()
    orr, _, hr, pr = rb.fetch()
    for j in range(len(seqs)):
        i = j * 50
        a, e = int(ou[i]), int(ou[i + 1])
        c, d = int(orr[2 * j]), int(orr[2 * j + 1])
        assert np.array_equal(hu[a:e], hr[c:d]) and np.array_equal(pu[a:e], pr[c:d])


def test_config4_shard_syncmer_k31_s11(engine, oracle):
    """BASELINE configs[3], ONE GPU's full shard of the 1 B-read job: 125M x 150 bp, closed syncmer k=31 s=11 (the other seven
    shards differ only in the seed; the 8-GPU run itself is the driver's)."""
    n = 125_000_000
    b = engine.synth(L.ALPHA_DNA, n, 150, SEED + 1)
    res = engine.run(b, engine.params(L.SYNCMER, 31, s=11))
    d = res.digest()
    assert 7.0 < d["n_tuples"] / n < 7.2  # measured density 7.09 per read
    assert d["short"] == 0 and d["illegal"] == 0 and d["has_non_acgt"] == 0
    assert 0.0005 < d["first_window_tie"] / n < 0.003  # ties that can reach the front of the first sorted window: 0.13 % of reads
    assert engine.run(b, engine.params(L.SYNCMER, 31, s=11), reuse=res).digest() == d  # deterministic
    # sampled bit-exact parity (tuples and flags) at the start, in the middle and at the very end of the shard
    for first in (0, 62_000_000, n - 800):
        data, offs = b.fetch_ascii(first, 800)
        o, st, h, p = res.fetch(first, 800)
        for r in range(800):
            eh, ep, es, fl = oracle.syncmer(data[int(offs[r]):int(offs[r + 1])].tobytes(), 31, 11, closed=True)
            a, e = int(o[r]), int(o[r + 1])
            assert np.array_equal(h[a:e], eh) and np.array_equal(p[a:e] & L.POS_MASK, ep) and np.array_equal(p[a:e] >> 31, es), (first, r)
            assert (int(st[r]) & 0xF0) == fl, (first, r)
    data, offs = b.fetch_ascii(0, 50_000)
    nt, ck = oracle.batch_run(5, data, offs, 31, 11, threads=os.cpu_count() or 1)
    small = engine.run(engine.synth(L.ALPHA_DNA, 50_000, 150, SEED + 1), engine.params(L.SYNCMER, 31, s=11)).digest()
    assert (small["n_tuples"], small["checksum"]) == (nt, ck)
    o, st, h, p = res.fetch(n - 3000, 3000)
    end = 150 - 2 * 31 + 11 + 1
    for r in range(3000):
        pos = p[int(o[r]):int(o[r + 1])] & L.POS_MASK
        assert np.all(np.diff(pos.astype(np.int64)) > 0) and (len(pos) == 0 or pos[-1] <= end)


def test_config5_protein_minimizer_k9_w5(engine, oracle):
    """BASELINE configs[4] at full size: 50M x 300 aa, protein minimizer k=9 w=5."""
    n = 50_000_000
    b = engine.synth(L.ALPHA_PROTEIN, n, 300, SEED + 2)
    res = engine.run(b, engine.params(L.PROT_MINIMIZER, 9, w=5))
    d = res.digest()
    assert 96.2 < d["n_tuples"] / n < 97.2  # 1 + 287*2/6 = 96.7
    assert d["short"] == 0 and d["n_tuples"] == res.info()["n_tuples"]
    assert engine.run(b, engine.params(L.PROT_MINIMIZER, 9, w=5), reuse=res).digest() == d
    for first in (0, 25_000_000, n - 300):  # sampled bit-exact parity: start, middle, end
        data, offs = b.fetch_ascii(first, 300)
        o, st, h, p = res.fetch(first, 300)
        for r in range(300):
            eh, ep, fl = oracle.protein_minimizer(data[int(offs[r]):int(offs[r + 1])].tobytes(), 9, 5)
            a, e = int(o[r]), int(o[r + 1])
            assert np.array_equal(h[a:e], eh) and np.array_equal(p[a:e] & L.POS_MASK, ep), (first, r)
            assert (int(st[r]) & 0xF0) == fl
    data, offs = b.fetch_ascii(0, 20_000)
    assert set(np.unique(data).tolist()) <= set(b"ACDEFGHIKLMNPQRSTVWY")
    nt, ck = oracle.batch_run(7, data, offs, 9, 5, threads=os.cpu_count() or 1)
    small = engine.run(engine.synth(L.ALPHA_PROTEIN, 20_000, 300, SEED + 2), engine.params(L.PROT_MINIMIZER, 9, w=5)).digest()
    assert (small["n_tuples"], small["checksum"]) == (nt, ck)
