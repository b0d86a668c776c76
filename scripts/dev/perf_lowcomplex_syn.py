"""dev (GPU): minimizer kernel on a batch where a fraction of the reads ends in a poly-A tail (key ties -> the exact list)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bio_amd import sketches as S, _lib as L
n, rl = 5_000_000, 150
rng = np.random.default_rng(3)
eng = S.Engine(0)
for frac in (0.0, 0.02, 0.10):
    data = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n * rl, dtype=np.uint8)].copy().reshape(n, rl)
    tail = rng.random(n) < frac
    data[tail, 100:] = ord("A")
    offs = np.arange(n + 1, dtype=np.uint64) * rl
    b = eng.batch_from_arrays(data.reshape(-1), offs)
    p = eng.params(L.SYNCMER, 31, s=11)
    res, ms = eng.run_timed(b, p, 1, 4)
    print("poly-A tails in %4.1f %% of the reads: %s  %.1f Gbases/s  tuples %d" % (100 * frac, res.plan()["kernel"], n * rl / min(ms) / 1e6, res.info()["n_tuples"]), flush=True)
