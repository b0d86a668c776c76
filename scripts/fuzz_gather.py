"""dev (GPU): differential fuzz of the group gathers (bsk_result_compact, bsk_result_fetch_narrow: k_gather_groups; bsk_result_sets'
k_move_groups through the per-sequence sets) against bsk_result_fetch's own wavefront-per-sequence gather, over random kinds, lengths,
batch sizes, ranges, low-complexity and empty reads.  usage: fuzz_gather.py first count"""
import ctypes as C
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bio_amd import _lib as L
from bio_amd import sketches as S

first, count = int(sys.argv[1]), int(sys.argv[2])
eng = S.Engine(0)
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]


def d2h(ptr, m, dt):
    a = np.empty(m, dt)
    if m:
        assert hip.hipMemcpy(a.ctypes.data, ptr, a.nbytes, 2) == 0
    return a


AA = "ACDEFGHIKLMNPQRSTVWY"
bad = 0
plans = {}
for seed in range(first, first + count):
    rng = random.Random(seed)
    kind = rng.choice(["min", "min", "min", "syn", "nt", "pmin", "phash", "kmer"])
    prot = kind in ("pmin", "phash")
    n = rng.choice([1, 63, 64, 65, 200, 1000, 4097, rng.randint(1, 3000)])
    shape = rng.choice(["short", "ragged", "mid", "long"])
    lens = {"short": lambda: rng.randint(100, 160), "ragged": lambda: rng.randint(0, 300), "mid": lambda: rng.randint(180, 400),
            "long": lambda: rng.choice([150, 150, 150, rng.randint(1000, 20000)])}[shape]
    if shape == "long":
        n = min(n, 300)
    alpha = AA if prot else "ACGT"
    seqs = []
    for _ in range(n):
        ln = lens()
        m = rng.random()
        if m < 0.03:
            seqs.append(alpha[0] * ln)
        elif m < 0.06:
            seqs.append((alpha[:2] * ln)[:ln])
        else:
            seqs.append("".join(rng.choice(alpha) for _ in range(ln)))
    try:
        b = eng.batch(seqs, L.ALPHA_PROTEIN if prot else L.ALPHA_DNA)
        if kind == "min":
            k, w = rng.choice([(21, 11), (15, 5), (31, 15), (11, 3), (25, 13)])
            p = eng.params(L.MINIMIZER, k, w=w)
        elif kind == "syn":
            k, s = rng.choice([(31, 11), (21, 10), (15, 8), (31, 16)])
            p = eng.params(L.SYNCMER, k, s=s)
        elif kind == "nt":
            p = eng.params(L.NTHASH, rng.choice([5, 21, 31]))
        elif kind == "kmer":
            p = eng.params(L.KMER, rng.choice([5, 21, 31]))
        elif kind == "pmin":
            k, w = rng.choice([(9, 5), (10, 3), (12, 8)])
            p = eng.params(L.PROT_MINIMIZER, k, w=w)
        else:
            p = eng.params(L.PROT_HASH, rng.choice([5, 9, 16]))
        res = eng.run(b, p)
        plans[res.plan()["kernel"].split("<")[0]] = plans.get(res.plan()["kernel"].split("<")[0], 0) + 1
        offs, st, h, pos = res.fetch()
        po, ph, pp, nt = res.compact()
        assert nt == int(offs[-1]) == (len(h) if nt else 0) or nt == 0, "count"
        assert np.array_equal(d2h(po, n + 1, np.uint64), offs), "compact offsets"
        assert np.array_equal(d2h(ph, nt, np.uint64), h[:nt]), "compact hashes"
        if pos is not None:
            assert np.array_equal(d2h(pp, nt, np.uint32), pos[:nt]), "compact positions"
        maxlen = max((len(q) for q in seqs), default=0)
        for _ in range(3):
            f = rng.randint(0, n - 1)
            c = rng.choice([n - f, rng.randint(0, n - f), min(n - f, 64), min(n - f, 1)])
            o, s1, hh, pp1 = res.fetch(f, c)
            if pos is not None and maxlen >= 32768:
                continue
            o2, s2, h2, p2 = res.fetch_narrow(f, c)
            T = int(o[-1])
            assert np.array_equal(o, o2.astype(np.uint64)) and np.array_equal(s1[:c], s2[:c]) and np.array_equal(hh[:T], h2[:T]), ("narrow", f, c)
            if pos is not None:
                assert np.array_equal(pp1[:T] & L.POS_MASK, (p2[:T] & 0x7FFF).astype(np.uint32)) and np.array_equal(pp1[:T] >> 31, (p2[:T] >> 15).astype(np.uint32)), ("narrow pos", f, c)
        if kind in ("min", "syn") and shape != "long":
            so, sv = res.sets(scale=rng.choice([1, 3]))
            os.environ["BSK_NO_GROUP_GATHER"] = "1"
            eng.reload_options()
            try:
                so2, sv2 = res.sets(scale=1)
                so3, sv3 = res.sets(scale=3)
            finally:
                del os.environ["BSK_NO_GROUP_GATHER"]
                eng.reload_options()
            assert (np.array_equal(so, so2) and np.array_equal(sv, sv2)) or (np.array_equal(so, so3) and np.array_equal(sv, sv3)), "sets"
        res.close()
        b.close()
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("SEED", seed, kind, shape, n, "FAILED:", repr(e)[:300], flush=True)
        if bad >= 5:
            break
print("done", count, "cases,", bad, "failures; plans:", plans)
