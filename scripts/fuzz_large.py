"""dev helper: random (kind, parameters) on large synthetic batches; device digest vs the oracle's batch driver.
usage: fuzz_large.py first count"""
import os, random, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bio_amd import sketches as S, _lib as L
from oracle import oracle as O

first, count = int(sys.argv[1]), int(sys.argv[2])
eng = S.Engine(0)
cores = len(os.sched_getaffinity(0))
bad = 0
for seed in range(first, first + count):
    rng = random.Random(seed)
    kind = rng.choice(["min", "min", "syn", "nt", "kmer", "sim", "pmin", "phash"])
    prot = kind in ("pmin", "phash")
    n = rng.choice([100_003, 250_000, 777_777, 1_200_000])
    length = rng.choice([60, 100, 150, 250]) if not prot else rng.choice([80, 300, 450])
    if kind == "min":
        k, x = rng.choice([11, 21, 31, 33]), rng.choice([2, 5, 8, 11, 15, 17, 24, 32, 40]); p = eng.params(L.MINIMIZER, k, w=x); ok = 4
    elif kind == "syn":
        k = rng.choice([21, 31]); x = rng.randint(max(1, k - 24), k - 1); p = eng.params(L.SYNCMER, k, s=x); ok = 5
    elif kind == "nt":
        k, x = rng.choice([5, 21, 31, 55]), 0; p = eng.params(L.NTHASH, k); ok = 2
    elif kind == "kmer":
        k, x = rng.choice([4, 21, 32]), 0; p = eng.params(L.KMER, k); ok = 1
    elif kind == "sim":
        k, x = rng.choice([21, 31]), 0; p = eng.params(L.SIMHASH, k, m=5, scale=5); ok = 3
    elif kind == "pmin":
        k, x = rng.choice([9, 10, 12]), rng.choice([3, 4, 5, 7]); p = eng.params(L.PROT_MINIMIZER, k, w=x); ok = 7
    else:
        k, x = rng.choice([5, 9, 12, 16, 20]), 0; p = eng.params(L.PROT_HASH, k); ok = 6
    if length < (3 * k + x if prot else 2 * k + x):
        length = (3 * k + x if prot else 2 * k + x) + 20
    t = time.time()
    b = eng.synth(L.ALPHA_PROTEIN if prot else L.ALPHA_DNA, n, length, seed)
    res = eng.run(b, p)
    d = res.digest()
    data, offs = b.fetch_ascii(0, n)
    nt, ck = O.batch_run(ok, data, offs, k, x, threads=cores)
    good = (nt, ck) == (d["n_tuples"], d["checksum"])
    print(f"seed {seed} {kind} k={k} x={x} n={n} L={length}: tuples {d['n_tuples']} {'ok' if good else 'MISMATCH oracle ' + str(nt)}  ({time.time()-t:.1f}s)", flush=True)
    bad += not good
    res.close(); b.close()
print("done", count, "cases,", bad, "failures")
