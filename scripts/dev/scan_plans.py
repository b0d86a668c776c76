"""dev (GPU): one rate per (kind, parameters, read length, batch flavour) off the tuned points -- what runs, how fast; looking for cliffs.
usage: python scripts/dev/scan_plans.py [bases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bio_amd import sketches as S, _lib as L

BASES = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5e9
eng = S.Engine(0)
rng = np.random.default_rng(3)


def host_batch(n, rl, frac_n=0.0, alpha=b"ACGT"):
    d = np.frombuffer(alpha, np.uint8)[rng.integers(0, len(alpha), n * rl, dtype=np.uint8)].copy()
    if frac_n:
        rows = np.nonzero(rng.random(n) < frac_n)[0]
        d[rows * rl + rng.integers(0, rl, len(rows))] = ord("N")
    return eng.batch_from_arrays(d, np.arange(n + 1, dtype=np.uint64) * rl)


CASES = []
for rl in (150, 1000):
    CASES += [("kmer k=21 canonical", dict(kind=L.KMER, k=21), rl, {}), ("kmer k=21 two strands", dict(kind=L.KMER, k=21, canonical=False), rl, {}),
              ("kmer k=32", dict(kind=L.KMER, k=32), rl, {}), ("nthash k=21 forward only", dict(kind=L.NTHASH, k=21, canonical=False), rl, {}),
              ("nthash k=64", dict(kind=L.NTHASH, k=64), rl, {}), ("nthash k=100", dict(kind=L.NTHASH, k=100), rl, {}),
              ("simhash k=31 m=7 scale=3", dict(kind=L.SIMHASH, k=31, m=7, scale=3), rl, {}), ("simhash k=16 m=4 scale=1", dict(kind=L.SIMHASH, k=16, m=4, scale=1), rl, {}),
              ("minimizer k=21 w=11 circular", dict(kind=L.MINIMIZER, k=21, w=11, circular=True), rl, {}),
              ("syncmer k=31 s=11 circular", dict(kind=L.SYNCMER, k=31, s=11, circular=True), rl, {}),
              ("minimizer k=21 w=11, 1 % reads with an N", dict(kind=L.MINIMIZER, k=21, w=11), rl, dict(frac_n=0.01)),
              ("minimizer k=21 w=11, 20 % reads with an N", dict(kind=L.MINIMIZER, k=21, w=11), rl, dict(frac_n=0.2)),
              ("syncmer k=31 s=11, 1 % reads with an N", dict(kind=L.SYNCMER, k=31, s=11), rl, dict(frac_n=0.01)),
              ("minimizer k=21 w=11, lower case + IUPAC in every read", dict(kind=L.MINIMIZER, k=21, w=11), rl, dict(alpha=b"ACGTacgtRY")),
              ("protein minimizer k=9 w=5 from DNA, frame 1", dict(kind=L.PROT_MINIMIZER, k=9, w=5, frame=1), rl, {}),
              ("protein minimizer k=7 w=3 from DNA, frame -2", dict(kind=L.PROT_MINIMIZER, k=7, w=3, frame=-2), rl, {}),
              ("protein hashes k=9 from DNA, frame 3", dict(kind=L.PROT_HASH, k=9, frame=3), rl, {}),
              ("minimizer k=64 w=11", dict(kind=L.MINIMIZER, k=64, w=11), rl, {}), ("minimizer k=100 w=11", dict(kind=L.MINIMIZER, k=100, w=11), rl, {}),
              ("syncmer k=64 s=48", dict(kind=L.SYNCMER, k=64, s=48), rl, {})]
for label, par, rl, bk in CASES:
    n = int(BASES / rl)
    try:
        b = host_batch(n, rl, **bk) if bk else eng.synth(L.ALPHA_DNA, n, rl, 0x5EED0003)
        kind = par.pop("kind")
        k = par.pop("k")
        res, ms = eng.run_timed(b, eng.params(kind, k, **par), 1, 3)
        par.update(kind=kind, k=k)
        print("%-58s %5d bp %8.1f Gbases/s  %s" % (label, rl, n * rl / min(ms) / 1e6, res.plan()["kernel"][:90]), flush=True)
        res.close()
        b.close()
    except Exception as e:
        print("%-58s %5d bp  ERROR %r" % (label, rl, e), flush=True)
