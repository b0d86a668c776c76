"""dev helper (GPU): differential fuzz of the class plans (run_classed, classes.hip) and of the pipeline's sink.

Every case: a batch of a random bulk length (fixed or ragged) with random outlier classes (counts, lengths up to 40 000 bases, reads with
an N, low-complexity reads), random (kind, k, w | s); the class-plan run (BSK_CLASS_FORCE, host list or device pass at random) must give,
read by read, what the one-plan run (BSK_NO_CLASS) gives -- hashes, positions, strands, status bytes -- and the same digest; every fourth
case also goes through the pipeline's sink (tuples, in order) and must equal the one-plan run.  usage: fuzz_class.py first count"""
import os, sys
os.environ.setdefault("BSK_PY_WATCH_ENV", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bio_amd import sketches as S, _lib as L

first, count = int(sys.argv[1]), int(sys.argv[2])
eng = S.Engine(0)
bad = 0


def case(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(17000, 40000))
    base = int(rng.choice([100, 150, 151, 200, 250]))
    ragged = rng.random() < 0.3
    lens = rng.integers(max(40, base - 60), base + 1, n).astype(np.uint64) if ragged else np.full(n, base, np.uint64)
    for _ in range(int(rng.integers(1, 5))):
        cnt = int(rng.choice([1, 3, 40, 400]))
        ln = int(rng.choice([30, 90, 260, 300, 420, 700, 1500, 5000, 9000, 40000]))
        spread = int(rng.integers(0, 30))
        idx = rng.integers(0, n, cnt)
        lens[idx] = rng.integers(max(1, ln - spread), ln + 1, cnt)
    offs = np.zeros(n + 1, np.uint64)
    np.cumsum(lens, out=offs[1:])
    data = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(offs[-1]), dtype=np.uint8)].copy()
    if rng.random() < 0.5:
        data[rng.integers(0, len(data), n // 60)] = ord("N")
    for r in rng.integers(0, n, 4):  # low-complexity reads
        data[int(offs[r]):int(offs[r + 1])] = ord("ACGT"[int(rng.integers(0, 4))])
    kind = L.MINIMIZER if rng.random() < 0.65 else L.SYNCMER
    if kind == L.MINIMIZER:
        k, w = int(rng.choice([15, 21, 31])), int(rng.choice([5, 9, 11, 13, 15]))
        p = eng.params(kind, k, w=w)
    else:
        k = int(rng.choice([25, 31]))
        p = eng.params(kind, k, s=int(rng.choice([11, 13])))
    b = eng.batch_from_arrays(data, offs)
    os.environ["BSK_NO_CLASS"] = "1"
    one = eng.run(b, p)
    del os.environ["BSK_NO_CLASS"]
    o_off, o_st, o_h, o_p = one.fetch()
    od = one.digest()
    os.environ["BSK_CLASS_FORCE"] = "1"
    view = rng.random() < 0.4
    if view:
        os.environ["BSK_CLASS_VIEW"] = "1"
    try:
        res = eng.run(b, p)
        plan = res.plan()["kernel"]
        c_off, c_st, c_h, c_p = res.fetch()
        assert res.digest() == od, ("digest", plan)
        assert np.array_equal(c_off, o_off) and np.array_equal(c_st[:n], o_st[:n]), ("offsets/status", plan)
        T = int(o_off[-1])
        assert np.array_equal(c_h[:T], o_h[:T]) and np.array_equal(c_p[:T], o_p[:T]), ("tuples", plan)
        res2, _ = eng.run_timed(b, p, 0, 1, reuse=res)
        assert res2.digest() == od, ("timed re-run", plan)
        if seed % 4 == 0:
            at = 0
            with S.Engine.pipeline_open(p, data=data, offsets=offs, devices=[0], n_streams=2, chunk_records=16500, sink=L.SINK_TUPLES, alphabet=L.ALPHA_DNA) as pl:
                for c in pl.chunks():
                    m = c.n_records
                    a0, a1 = int(o_off[at]), int(o_off[at + m])
                    assert np.array_equal(c.hash, o_h[a0:a1]) and np.array_equal(c.status, o_st[at:at + m]), ("sink", at)
                    if c.pos is not None:
                        assert np.array_equal(c.pos, o_p[a0:a1] & L.POS_MASK), ("sink pos", at)
                    at += m
            assert at == n
        res.close()
    finally:
        del os.environ["BSK_CLASS_FORCE"]
        os.environ.pop("BSK_CLASS_VIEW", None)
    one.close()
    b.close()
    return plan


parts = 0
for seed in range(first, first + count):
    try:
        pl = case(seed)
        parts += " reads of " in pl
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("SEED", seed, "FAILED:", repr(e)[:300], flush=True)
        if bad >= 5:
            break
print("done", count, "cases,", parts, "with a class plan,", bad, "failures")
