"""Seeded differential fuzz: random kinds, parameters, length distributions, alphabets and plan switches against the oracle.

Every case draws its own batch shape (short reads, ragged reads, reads beyond the 512-base stream limit, contigs beyond the
tile threshold, empty reads), its own letters (pure ACGT, a few N, IUPAC + lower case, junk letters) and sometimes forces
small tiles or the general kernels, so that plan boundaries (fast / general / mixed / tiled) are crossed in combinations
the hand-written tests do not list.
"""
import os
import random

import numpy as np
import pytest

from bio_amd import _lib as L

pytestmark = pytest.mark.gpu

AA = "ACDEFGHIKLMNPQRSTVWY"


def draw_batch(rng, protein=False):
    shape = rng.choice(["reads", "ragged", "long", "mixedlen", "tiny", "many"])
    n = rng.randint(1, 180) if shape != "many" else rng.randint(500, 2500)  # "many": several tickets of 8 units
    lens = []
    for _ in range(n):
        if shape == "reads":
            lens.append(rng.choice([100, 150, 151, 250]))
        elif shape == "ragged":
            lens.append(rng.randint(0, 400))
        elif shape == "long":
            lens.append(rng.choice([rng.randint(400, 900), rng.randint(900, 6000)]))
        elif shape == "mixedlen":
            lens.append(rng.choice([0, 1, 30, 150, 513, 700, 4097, rng.randint(1, 9000)]))
        elif shape == "many":
            lens.append(rng.choice([rng.randint(20, 220), 150]))
        else:
            lens.append(rng.randint(0, 40))
    if protein:
        alpha = rng.choice([AA, AA, AA + "X*", "AC"])
        seqs = ["".join(rng.choice(alpha) for _ in range(min(x, 1500))) for x in lens]
        if rng.random() < 0.3:  # a low-complexity sequence among them (dense minimizers: slab overflow of the register kernel)
            seqs[rng.randrange(len(seqs))] = rng.choice(["A", "AC", "ACD"]) * rng.randint(10, 200)
        return seqs
    style = rng.choice(["acgt", "acgt", "fewN", "iupac", "junk"])
    seqs = []
    for x in lens:
        s = [rng.choice("ACGT") for _ in range(x)]
        if style == "fewN" and x and rng.random() < 0.15:
            for _ in range(rng.randint(1, 3)):
                s[rng.randrange(x)] = "N"
        elif style == "iupac" and x and rng.random() < 0.5:
            for _ in range(rng.randint(1, max(1, x // 10))):
                s[rng.randrange(x)] = rng.choice("acgtnNRYKMSWBDHVU")
        elif style == "junk" and x and rng.random() < 0.2:
            s[rng.randrange(x)] = rng.choice("X-*. 7")
        seqs.append("".join(s))
    return seqs


def env_switches(rng):
    env = {}
    r = rng.random()
    if r < 0.25:
        env["BSK_TILE_MIN"] = str(rng.choice([40, 64, 200]))
        env["BSK_TILE_POS"] = str(rng.choice([16, 32, 48, 128]))
    elif r < 0.35:
        env["BSK_FORCE_GENERIC"] = "1"
    elif r < 0.45:
        env["BSK_NO_MIXED"] = "1"
    elif r < 0.5:
        env["BSK_NO_TILES"] = "1"
    if rng.random() < 0.5:
        env["BSK_BIN_MIN"] = "1"  # length-binned units on these small batches too (round 4)
    return env


def run_case(engine, oracle, seed):
    rng = random.Random(seed)
    kind = rng.choice([L.MINIMIZER, L.MINIMIZER, L.SYNCMER, L.NTHASH, L.KMER, L.SIMHASH, L.PROT_HASH, L.PROT_MINIMIZER])
    env = env_switches(rng)
    old = {k: os.environ.get(k) for k in ("BSK_TILE_MIN", "BSK_TILE_POS", "BSK_FORCE_GENERIC", "BSK_NO_MIXED", "BSK_NO_TILES", "BSK_BIN_MIN")}
    for k in old:
        os.environ.pop(k, None)
    os.environ.update(env)
    try:
        protein = kind in (L.PROT_HASH, L.PROT_MINIMIZER)
        dna_fed = protein and rng.random() < 0.4
        seqs = draw_batch(rng, protein and not dna_fed)
        circular = (not protein) and rng.random() < 0.15
        if kind == L.MINIMIZER:
            k, w = rng.choice([5, 11, 15, 21, 31, 33, 64]), rng.choice([1, 2, 3, 5, 7, 11, 12, 16, 17, 20, 32, 40])
            pk, fn = dict(k=k, w=w, circular=circular), lambda q: oracle.minimizer(q, k, w, circular, closed=True)
        elif kind == L.SYNCMER:
            k = rng.choice([7, 15, 21, 31, 40])
            s = rng.randint(1, k)
            pk, fn = dict(k=k, s=s, circular=circular), lambda q: oracle.syncmer(q, k, s, circular, closed=True)
        elif kind == L.NTHASH:
            k, canon = rng.choice([1, 5, 21, 31, 64, 100, 300]), rng.random() < 0.7
            pk, fn = dict(k=k, canonical=canon, circular=circular), lambda q: (oracle.nthash(q, k, canon, circular)[0], None, None, None)
        elif kind == L.KMER:
            k, canon = rng.choice([1, 4, 11, 21, 31, 32]), rng.random() < 0.6
            nuc_alpha = rng.choice([L.ALPHA_DNA, L.ALPHA_DNA, L.ALPHA_DNA_PLAIN, L.ALPHA_RNA, L.ALPHA_RNA_REDUNDANT, L.ALPHA_UNLIMIT])  # second strand: PairLetter of the Seq's alphabet
            pk, fn = dict(k=k, canonical=canon, circular=circular), lambda q: (oracle.kmer_codes(q, k, canon, circular, nuc_alpha), None, None, None)
        elif kind == L.SIMHASH:
            k = rng.choice([8, 16, 21, 31, 40, 70])
            m = rng.randint(4, min(k, 12))
            scale = rng.randint(1, min(k - m + 1, 9))
            canon = rng.random() < 0.7
            pk, fn = dict(k=k, m=m, scale=scale, canonical=canon), lambda q: (oracle.simhash(q, k, m, scale, canon), None, None, None)
        elif kind == L.PROT_HASH:
            k = rng.choice([2, 5, 9, 10, 12, 16, 17, 33])
            table, frame = rng.choice([1, 2, 4, 11]), rng.choice([1, 2, 3, -1, -2, -3])
            pk = dict(k=k, codon_table=table, frame=frame)
            fn = (lambda q: (oracle.protein_hashes_nt(q, k, table, frame), None, None, None)) if dna_fed else \
                (lambda q: (oracle.protein_hashes(q, k), None, None, None))
        else:
            k, w = rng.choice([3, 9, 10, 12, 14]), rng.choice([1, 3, 4, 5, 8])
            table, frame = rng.choice([1, 11]), rng.choice([1, 3, -2])
            pk = dict(k=k, w=w, codon_table=table, frame=frame)
            fn = (lambda q: oracle.protein_minimizer_nt(q, k, w, table, frame)[:2] + (None, None)) if dna_fed else \
                (lambda q: oracle.protein_minimizer(q, k, w, closed=True)[:2] + (None, None))
        b = engine.batch(seqs, L.ALPHA_PROTEIN if (protein and not dna_fed) else (nuc_alpha if kind == L.KMER else L.ALPHA_DNA))
        try:
            res = engine.run(b, engine.params(kind, **pk))
        except Exception as e:  # refusals must be the documented ones
            msg = str(e)
            assert "2^24" in msg or "unsupported" in msg.lower(), (seed, kind, pk, msg)
            return
        for i, q in enumerate(seqs):
            st, h, p = res.read(i)
            try:
                eh, ep, es, _ = fn(q)
            except oracle.OracleError as e:
                if e.name == "ErrShortSeq":
                    assert (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0, (seed, kind, pk, env, i, len(q))
                else:
                    assert e.name == "ErrIllegalBase" and (st & L.ST_CODE_MASK) == L.ST_ILLEGAL, (seed, kind, pk, env, i, e.name, st)
                continue
            assert (st & L.ST_CODE_MASK) == L.ST_OK, (seed, kind, pk, env, i, len(q), st)
            assert len(h) == len(eh) and np.array_equal(h, eh), (seed, kind, pk, env, i, len(q), len(h), len(eh))
            if ep is not None:
                assert np.array_equal(p & L.POS_MASK, ep), (seed, kind, pk, env, i)
            if es is not None:
                assert np.array_equal(p >> 31, es), (seed, kind, pk, env, i)
        d = res.digest()
        assert d["n_tuples"] == res.info()["n_tuples"], (seed, kind, pk, env)
        if rng.random() < 0.3:  # sketch sets of the same result against numpy
            whole, scale = rng.random() < 0.4, rng.choice([1, 1, 2, 9])
            offs, vals = res.sets(whole_batch=whole, scale=scale)
            maxhash = (2**64 - 1) // scale if scale > 1 else 2**64 - 1
            per = []
            for i in range(len(seqs)):
                h = res.read(i)[1]
                per.append(np.unique(h[h <= np.uint64(maxhash)]))
            want = [np.unique(np.concatenate(per))] if whole else per
            assert len(offs) == len(want) + 1, (seed, kind, pk, env)
            for i, w_ in enumerate(want):
                assert np.array_equal(vals[int(offs[i]):int(offs[i + 1])], w_), (seed, kind, pk, env, "sets", whole, scale, i)
        if not protein and max((len(q) for q in seqs), default=0) < 5000 and rng.random() < 0.2:  # translation of the same batch
            table, frame = rng.choice([1, 3, 11, 25]), rng.choice([1, 2, 3, -1, -2, -3])
            t = b.translate(table, frame)
            data, toffs = t.fetch_ascii(0, len(seqs))
            for i, q in enumerate(seqs):
                got = data[int(toffs[i]):int(toffs[i + 1])].tobytes().decode("latin-1")
                want = oracle.translate(q, table, frame) if len(q) >= 3 else ""
                assert got == want, (seed, "translate", table, frame, i, len(q))
            t.close()
        res.close()
        b.close()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("block", range(12))
def test_differential_fuzz(engine, oracle, block):
    for seed in range(block * 25, block * 25 + 25):
        run_case(engine, oracle, 0xF0220000 + seed)


def test_regressions_found_by_the_fuzz_campaign(engine, oracle):
    """(1) s == k syncmers of reads with few k-mers took the staged path without staged positions (GPU memory fault);
    (2) a tiled run whose batch has no tile at all read stale device counters (bogus capacity, out of memory)."""
    rng = random.Random(1)
    seqs = ["".join(rng.choice("ACGT") for _ in range(n)) for n in (31, 32, 35, 40, 40, 30, 12, 0)]
    b = engine.batch(seqs)
    res = engine.run(b, engine.params(L.SYNCMER, 31, s=31))
    for i, q in enumerate(seqs):
        st, h, p = res.read(i)
        try:
            eh, ep, es, _ = oracle.syncmer(q, 31, 31, False, closed=True)
        except oracle.OracleError:
            assert (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0
            continue
        assert np.array_equal(h, eh) and np.array_equal(p & L.POS_MASK, ep)
    os.environ["BSK_TILE_MIN"] = "40"
    try:
        for kind, pk in ((L.MINIMIZER, dict(k=31, w=16, circular=True)), (L.SYNCMER, dict(k=40, s=19, circular=True)),
                         (L.NTHASH, dict(k=41, circular=True))):
            res = engine.run(b, engine.params(kind, **pk))  # every read is too short: a tiled run without tiles
            assert res.info()["n_tuples"] == 0 and all((res.read(i)[0] & L.ST_CODE_MASK) == L.ST_SHORT for i in range(len(seqs)))
    finally:
        del os.environ["BSK_TILE_MIN"]
