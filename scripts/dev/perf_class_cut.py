"""dev (GPU): what cutting a batch with MANY odd reads costs (10^8 x 150 bases + 1 % of 250): bsk_batch_prepare's time, rate, digests."""
import os, sys, json
os.environ.setdefault("BSK_PY_WATCH_ENV", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bio_amd import sketches as S, _lib as L
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
eng = S.Engine(0)
rng = np.random.default_rng(12)
lens = np.full(n, 150, np.uint64)
lens[rng.integers(0, n, n // 100)] = 250
lens[rng.integers(0, n, n // 10000)] = 400
offs = np.zeros(n + 1, np.uint64)
np.cumsum(lens, out=offs[1:])
data = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(offs[-1]), dtype=np.uint8)]
b = eng.batch_from_arrays(data, offs)
p = eng.params(L.MINIMIZER, 21, w=11)
res, ms = eng.run_timed(b, p, 2, 5)
prep = sorted(eng.prepare(b, p) for _ in range(5))
print("class plans: %.1f Gbases/s, cut %.3f ms (min of 5; median %.3f), %s" % (int(offs[-1]) / min(ms) / 1e6, prep[0], prep[2], res.plan()["kernel"][:150]))
d = res.digest()["checksum"]
res.close()
os.environ["BSK_NO_CLASS"] = "1"
res, ms = eng.run_timed(b, p, 1, 3)
print("one plan: %.1f Gbases/s, digests equal: %s" % (int(offs[-1]) / min(ms) / 1e6, res.digest()["checksum"] == d))
