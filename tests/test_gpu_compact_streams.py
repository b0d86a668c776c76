"""GPU: k_nthash_fast<MODE, true> -- fixed-length batches leave without padding (runs back to back inside a unit, the line two reads
share assembled at the end of the unit).  Every value of every read against the oracle for read lengths that put the reads' starts at
every offset inside a line, partial last units, the three modes (forward ntHash, canonical ntHash, canonical k-mer codes), and the same
digest / fetch as the line-padded kernel (BSK_NO_COMPACT).  Contract: iterator.go:658-665 (NextHash), :708-759 (NextKmer)."""
import random

import numpy as np
import pytest

from bio_amd import _lib as L

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _compact_build(engine, monkeypatch):
    """The compact-run kernels are an experiment (measured and rejected: fewer bytes, slower): make EXPERIMENTS=1 + BSK_COMPACT=1."""
    if not engine.lib.bsk_build_has_experiments():
        pytest.skip("library built without EXPERIMENTS=1")
    monkeypatch.setenv("BSK_COMPACT", "1")


def rand_dna(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


@pytest.mark.parametrize("k,rl,n", [(21, 150, 1000), (21, 151, 777), (21, 52, 300), (21, 53, 64), (31, 100, 129), (15, 77, 640), (21, 250, 200),
                                    (5, 36, 500), (21, 500, 70), (64, 95, 100), (21, 51, 100)])
def test_compact_runs_every_value(engine, oracle, k, rl, n):
    rng = random.Random(k * 1000 + rl)
    seqs = [rand_dna(rng, rl) for _ in range(n)]
    b = engine.batch(seqs)
    for canonical in (True, False):
        res = engine.run(b, engine.params(L.NTHASH, k, canonical=canonical))
        want = rl >= k + 31
        assert ("true>" in res.plan()["kernel"]) == want, res.plan()
        for i, s in enumerate(seqs):
            st, h, _ = res.read(i)
            eh, _es = oracle.nthash(s, k, canonical, False)
            assert (st & L.ST_CODE_MASK) == L.ST_OK and np.array_equal(h, eh), (i, k, rl, canonical)
        res.close()
    if k <= 32:
        res = engine.run(b, engine.params(L.KMER, k, canonical=True))
        assert ("true>" in res.plan()["kernel"]) == (rl >= k + 31), res.plan()
        for i, s in enumerate(seqs):
            st, h, _ = res.read(i)
            assert np.array_equal(h, oracle.kmer_codes(s, k, True, False)), (i, k, rl)
        res.close()
    b.close()


def test_compact_equals_padded(engine, monkeypatch):
    b = engine.synth(L.ALPHA_DNA, 300_001, 150, 0x5EED0003)
    p = engine.params(L.NTHASH, 21)
    res = engine.run(b, p)
    assert "k_nthash_fast<1,true>" in res.plan()["kernel"]
    d1, f1 = res.digest(), res.fetch(299_000, 1001)
    inf = res.info()
    assert inf["n_tuples"] == 300_001 * 130
    res.close()
    monkeypatch.delenv("BSK_COMPACT")
    res = engine.run(b, p)
    assert "k_nthash_fast<1>" in res.plan()["kernel"]
    d2, f2 = res.digest(), res.fetch(299_000, 1001)
    res.close()
    b.close()
    assert d1 == d2
    assert all(np.array_equal(x, y) for x, y in zip(f1[:3], f2[:3]))


def test_compact_runs_are_not_used_for_mixed_or_ragged_batches(engine, oracle):
    rng = random.Random(3)
    seqs = [rand_dna(rng, 150) for _ in range(300)]
    seqs[17] = seqs[17][:40] + "N" + seqs[17][41:]
    b = engine.batch(seqs)
    res = engine.run(b, engine.params(L.NTHASH, 21))
    assert "true>" not in res.plan()["kernel"], res.plan()
    for i in (0, 16, 17, 18, 299):
        assert np.array_equal(res.read(i)[1], oracle.nthash(seqs[i], 21, True, False)[0])
    res.close()
    b.close()
