"""dev helper: a batch where 1 % of the reads carry an N -- mixed plan vs whole-batch ASCII kernels."""
import os, sys, time
os.environ.setdefault("BSK_PY_WATCH_ENV", "1")  # this script flips BSK_* switches between runs
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bio_amd import sketches as S, _lib as L
n, length = int(float(sys.argv[1])) if len(sys.argv) > 1 else 5_000_000, 150
rng = np.random.default_rng(3)
data = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n * length, dtype=np.uint8)].copy()
bad = rng.choice(n, n // 100, replace=False)
data[bad * length + rng.integers(0, length, bad.size)] = ord("N")
offs = np.arange(n + 1, dtype=np.uint64) * length
eng = S.Engine(0)
b = eng.batch_from_arrays(data, offs)
print(b.info())
for name, p in (("minimizer k21 w11", eng.params(L.MINIMIZER, 21, w=11)), ("nthash k21", eng.params(L.NTHASH, 21)), ("syncmer k31 s11", eng.params(L.SYNCMER, 31, s=11))):
    for mode in ("mixed", "ascii-all"):
        if mode == "ascii-all": os.environ["BSK_NO_MIXED"] = "1"
        else: os.environ.pop("BSK_NO_MIXED", None)
        res, ms = eng.run_timed(b, p, 1, 3)
        print(f"{name:18s} {mode:10s} kernel ms {[round(m,3) for m in ms]} -> {n*length/min(ms)/1e6:.0f} Gbases/s  digest {res.digest()['checksum']}")
        res.close()
