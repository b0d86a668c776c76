"""dev (GPU): minimizer kernel on fixed-length against ragged batches of the same size (trimmed reads: lengths uniform in [lo, 150])."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bio_amd import sketches as S, _lib as L
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
rng = np.random.default_rng(5)
eng = S.Engine(0)
p = eng.params(L.MINIMIZER, 21, w=11)
lows = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (150, 140, 100, 60)
for lo in lows:
    lens = rng.integers(lo, 151, n, dtype=np.uint64)
    offs = np.zeros(n + 1, np.uint64)
    np.cumsum(lens, out=offs[1:])
    data = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(offs[-1]), dtype=np.uint8)]
    b = eng.batch_from_arrays(data, offs)
    res, ms = eng.run_timed(b, p, 1, 4)
    print("lengths %d..150: %s  %.1f Gbases/s  (%.3f ms, %d tuples)" % (lo, res.plan()["kernel"], int(offs[-1]) / min(ms) / 1e6, min(ms), res.info()["n_tuples"]), flush=True)
