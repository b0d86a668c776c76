"""dev (GPU): batches with an N in some reads (2-bit kernel over everything + the ASCII side launch): beside the main kernel / behind it"""
import os, sys
os.environ.setdefault("BSK_PY_WATCH_ENV", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bio_amd import sketches as S, _lib as L
eng = S.Engine(0)
rng = np.random.default_rng(3)
for rl in (150, 1000):
    n = int(1.5e9 / rl)
    for frac in (0.0, 0.001, 0.01, 0.05, 0.2):
        d = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n * rl, dtype=np.uint8)].copy()
        rows = np.nonzero(rng.random(n) < frac)[0]
        d[rows * rl + rng.integers(0, rl, len(rows))] = ord("N")
        b = eng.batch_from_arrays(d, np.arange(n + 1, dtype=np.uint64) * rl)
        for name, p in (("min", eng.params(L.MINIMIZER, 21, w=11)), ("syn", eng.params(L.SYNCMER, 31, s=11))):
            row = []
            dg = []
            for mode, env in (("beside", {}), ("behind", {"BSK_NO_SIDE_EARLY": "1"})):
                for k_, v in env.items(): os.environ[k_] = v
                res, ms = eng.run_timed(b, p, 1, 4)
                row.append("%s %7.1f" % (mode, n * rl / min(ms) / 1e6))
                dg.append(res.digest()["checksum"])
                pl = res.plan()["kernel"]
                res.close()
                for k_ in env: del os.environ[k_]
            print("%4d bp %5.1f %% with an N  %s: %s  %s  %s" % (rl, frac * 100, name, " | ".join(row), "digests equal" if len(set(dg)) == 1 else "DIGESTS DIFFER", pl[:60]), flush=True)
        b.close()
