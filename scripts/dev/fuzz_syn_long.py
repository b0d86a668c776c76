"""dev (GPU): differential fuzz aimed at k_syncmer_pkl (round 4): k - s = 4..24, s >= 9, ragged batches of reads up to 480 bases with
low-complexity reads mixed in, length-binned units on and off; every read against the oracle's closed form, every fourth against the
reference's state machine.  usage: fuzz_syn_long.py first count"""
import os, sys, random
os.environ.setdefault("BSK_PY_WATCH_ENV", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from bio_amd import sketches as S, _lib as L
from oracle import oracle
first, count = int(sys.argv[1]), int(sys.argv[2])
eng = S.Engine(0)
bad = 0
plans = {}
for seed in range(first, first + count):
    rng = random.Random(seed)
    w = rng.randint(4, 24)
    s = rng.randint(9, 24)
    k = s + w
    hi = rng.choice([150, 200, 250, 300, 352, 400, 480])
    lo = rng.choice([max(2 * k - s - 1, hi - 60), hi - 1, max(40, hi // 3)])
    n = rng.choice([64, 200, 700, 4500])
    seqs = []
    for _ in range(n):
        Ln = rng.randint(min(lo, hi), hi)
        r = rng.random()
        if r < 0.02:
            q = rng.choice("ACGT") * Ln
        elif r < 0.05:
            unit = "".join(rng.choice("ACGT") for _ in range(rng.randint(2, 7)))
            q = (unit * (Ln // len(unit) + 1))[:Ln]
        elif r < 0.08:
            t = rng.randint(20, 60)
            q = "".join(rng.choice("ACGT") for _ in range(max(Ln - t, 0))) + rng.choice("AG") * min(t, Ln)
        else:
            q = "".join(rng.choice("ACGT") for _ in range(Ln))
        seqs.append(q)
    os.environ.pop("BSK_BIN_MIN", None)
    if rng.random() < 0.5:
        os.environ["BSK_BIN_MIN"] = "1"
    eng.reload_options()
    b = eng.batch(seqs)
    res = eng.run(b, eng.params(L.SYNCMER, k, s=s))
    kern = res.plan()["kernel"].split("<")[0]
    plans[kern] = plans.get(kern, 0) + 1
    try:
        for i, q in enumerate(seqs):
            st, h, p = res.read(i)
            if len(q) < 2 * k - s - 1:
                assert (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0, (i, len(q))
                continue
            eh, ep, es, fl = oracle.syncmer(q, k, s, False, closed=True)
            assert (st & L.ST_CODE_MASK) == L.ST_OK and np.array_equal(h, eh), (i, len(q), "hash")
            assert np.array_equal(p & L.POS_MASK, ep) and np.array_equal(p >> 31, es) and (st & 0xF0) == fl, (i, len(q), "pos/strand/flag")
            if i % 4 == 0:
                mh, mp, _, _ = oracle.syncmer(q, k, s)
                assert np.array_equal(h, mh) and np.array_equal(p & L.POS_MASK, mp), (i, len(q), "state machine")
    except AssertionError as e:
        bad += 1
        print("SEED", seed, "k", k, "s", s, "hi", hi, "n", n, res.plan(), "FAILED:", repr(e)[:300], flush=True)
        if bad >= 5:
            break
    res.close()
    b.close()
print("done", count, "cases,", bad, "failures; plans:", plans)
