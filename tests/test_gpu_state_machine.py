"""GPU parity against the LINE-BY-LINE restatement of the reference's state machines, not only against the closed forms.

The kernels implement closed forms (leftmost minimum of every window, once per position); `tests/test_oracle_golden.py` shows closed
form == state machine on the CPU.  Here the HIP path itself is held against the restated machines -- the sorted first window, the
binary-search insert / delete of `sketch.go:263-295`, the protein copy of it (`sketch-protein.go:106-210`) -- on every kernel family
a batch can be planned on (round 3's review: "only syncmers are also checked against the state machine on the GPU").
Contract: NextMinimizer `sketches/sketch.go:205-309`, ProteinMinimizerSketch.Next `sketches/sketch-protein.go:106-210`.
"""
import random

import numpy as np
import pytest

from bio_amd import _lib as L

pytestmark = pytest.mark.gpu

AA = "ACDEFGHIKLMNPQRSTVWY"


def rand_seq(rng, n, alpha="ACGT"):
    return "".join(rng.choice(alpha) for _ in range(n))


def low_complexity(rng, n):
    """repeats and homopolymer runs: equal hashes inside a window, the case where an unstable buffer order could show"""
    unit = rand_seq(rng, rng.randint(1, 6))
    s = (unit * (n // len(unit) + 1))[:n]
    cut = rng.randint(0, n)
    return s[:cut] + rand_seq(rng, n - cut)


# (k, w, lengths, kernel the planner must name) -- one row per kernel family of the minimizer path
CASES = [
    (21, 11, (150,), "k_minimizer_pk"),             # headline plan: packed 32-bit window machine
    (21, 11, (60, 100, 150), "k_minimizer_pk"),     # ragged (length-binned units when the batch is large enough)
    (21, 11, (250,), "k_minimizer_ring"),           # unit rows
    (21, 11, (200, 260, 290), "k_minimizer_ring"),
    (21, 11, (200, 300, 350), "k_minimizer_pkd"),   # the packed machine over per-read slabs and mid-read flushes (round 5)
    (21, 11, (900, 1700), "k_minimizer_pkd"),
    (15, 5, (300, 420), "k_minimizer_pkd"),         # two blocks per flush round
    (15, 3, (260,), "k_minimizer_pkd"),             # four blocks per flush round
    (21, 13, (700,), "k_minimizer_pkd"),
    (31, 15, (150,), "k_minimizer_fast"),           # the reference's own benchmark parameters (sketch_test.go:128), w >= 14
    (21, 5, (150,), "k_minimizer_"),               # dense selection (small w): unit rows or k_minimizer_dense
    (21, 11, (500, 700), "k_minimizer_pkd"),
    (15, 8, (5000, 9000), "over tiles"),                 # long sequences as tiles
    (64, 20, (150, 220), "minimizer"),            # k = 64: the rotation's last step
]


@pytest.mark.parametrize("k,w,lens,kernel", CASES)
def test_minimizer_matches_state_machine(engine, oracle, k, w, lens, kernel):
    rng = random.Random(131 * k + w + len(lens))
    n = 1500 if max(lens) <= 400 else 60
    seqs = [rand_seq(rng, rng.choice(lens)) for _ in range(n)]
    for j in range(0, n, 10):                       # every tenth read is low-complexity
        seqs[j] = low_complexity(rng, len(seqs[j]))
    seqs[1] = "A" * len(seqs[1])
    b = engine.batch(seqs)
    res = engine.run(b, engine.params(L.MINIMIZER, k, w=w))
    assert kernel in res.plan()["kernel"], res.plan()
    flagged = 0
    for i, q in enumerate(seqs):
        st, h, p = res.read(i)
        try:
            mh, mp, ms, _ = oracle.minimizer(q, k, w)   # closed=False: the state machine
        except oracle.OracleError as e:
            assert e.name == "ErrShortSeq" and (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0, (i, len(q))
            continue
        assert (st & L.ST_CODE_MASK) == L.ST_OK, (i, len(q))
        flagged += bool(st & L.ST_FIRST_WINDOW_TIE)
        assert np.array_equal(h, mh), (i, k, w, len(q))
        assert np.array_equal(p & L.POS_MASK, mp) and np.array_equal(p >> 31, ms), (i, k, w, len(q))
    assert flagged >= 1                             # the poly-A read at least: ties are in the sample
    res.close()
    b.close()


@pytest.mark.parametrize("k,w,lens", [(9, 5, (300,)), (9, 5, (40, 120, 300, 700)), (12, 8, (300,)), (16, 2, (200,)), (10, 3, (5000,)), (33, 4, (300,))])
def test_protein_minimizer_matches_state_machine(engine, oracle, k, w, lens):
    rng = random.Random(977 * k + w)
    n = 600 if max(lens) <= 1000 else 40
    seqs = [rand_seq(rng, rng.choice(lens), AA) for _ in range(n)]
    for j in range(0, n, 10):
        seqs[j] = low_complexity(rng, len(seqs[j])).replace("T", "W")
    b = engine.batch(seqs, L.ALPHA_PROTEIN)
    res = engine.run(b, engine.params(L.PROT_MINIMIZER, k, w=w))
    for i, q in enumerate(seqs):
        st, h, p = res.read(i)
        try:
            mh, mp, _ = oracle.protein_minimizer(q, k, w)  # the state machine
        except oracle.OracleError as e:
            assert e.name == "ErrShortSeq" and (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0, (i, len(q))
            continue
        assert (st & L.ST_CODE_MASK) == L.ST_OK and np.array_equal(h, mh) and np.array_equal(p & L.POS_MASK, mp), (i, k, w, len(q))
    res.close()
    b.close()


@pytest.mark.parametrize("frame", [1, -2])
def test_dna_fed_protein_minimizer_matches_state_machine(engine, oracle, frame):
    """2-bit DNA in, translated inside the kernel (kernels_translate.hpp), against translate + the protein state machine."""
    rng = random.Random(4100 + frame)
    seqs = [rand_seq(rng, rng.choice((150, 300, 451))) for _ in range(600)]
    b = engine.batch(seqs)
    res = engine.run(b, engine.params(L.PROT_MINIMIZER, 9, w=5, codon_table=1, frame=frame))
    for i, q in enumerate(seqs):
        st, h, p = res.read(i)
        aa = oracle.translate(q, 1, frame)
        try:
            mh, mp, _ = oracle.protein_minimizer(aa, 9, 5)
        except oracle.OracleError:
            continue
        if len(q) < 9 * 3 + 5 - 1:
            continue
        assert (st & L.ST_CODE_MASK) == L.ST_OK and np.array_equal(h, mh) and np.array_equal(p & L.POS_MASK, mp), (i, frame, len(q))
    res.close()
    b.close()


@pytest.mark.parametrize("k,w", [(21, 11), (15, 5)])
def test_pkd_batches_of_low_complexity_reads(engine, oracle, k, w):
    """k_minimizer_pkd when MOST reads are the exact machine's: homopolymers (every position selected: the read outgrows its slab and its
    ring, `lost`), dinucleotide repeats (key ties in every window), half a batch of poly-A -- the list of reads, its slabs in the overflow
    region and the first-window flag, read by read against the closed form"""
    rng = random.Random(5 * k + w)
    makers = (lambda i: "ACGT"[i % 4] * 400, lambda i: ("ACGTAG"[i % 5:i % 5 + 2] * 300)[:500],
              lambda i: ("A" * 450) if i % 2 else rand_seq(rng, 450))
    for mk in makers:
        seqs = [mk(i) for i in range(4000)]
        b = engine.batch(seqs)
        res = engine.run(b, engine.params(L.MINIMIZER, k, w=w))
        assert "k_minimizer_pkd" in res.plan()["kernel"], res.plan()
        for i in list(range(0, 4000, 211)) + [1, 2, 3]:
            st, h, p = res.read(i)
            eh, ep, es, fl = oracle.minimizer(seqs[i], k, w, False, closed=True)
            assert np.array_equal(h, eh) and np.array_equal(p & L.POS_MASK, ep) and np.array_equal(p >> 31, es), (i, len(seqs[i]))
            assert bool(st & L.ST_FIRST_WINDOW_TIE) == bool(fl & oracle.FLAG_FIRST_WINDOW_TIE), i
        res.close()
        b.close()
