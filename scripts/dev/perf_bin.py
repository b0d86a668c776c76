"""dev (GPU): the length-binned view built with the batch (bin_with_batch) against the per-plan pass (BSK_NO_BIN_EARLY=1) and batch order
(BSK_NO_BIN=1) on ragged 60..150-base reads: kernel rate, the pass's own time, digests.  usage: python scripts/dev/perf_bin.py [reads]"""
import os, sys, time
os.environ.setdefault("BSK_PY_WATCH_ENV", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bio_amd import sketches as S, _lib as L

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 19_000_000
eng = S.Engine(0)
rng = np.random.default_rng(12)
lens = rng.integers(60, 151, n, dtype=np.uint64)
offs = np.zeros(n + 1, np.uint64)
np.cumsum(lens, out=offs[1:])
data = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(offs[-1]), dtype=np.uint8)]
nb = int(offs[-1])
u = eng.synth(L.ALPHA_DNA, int(nb / 150), 150, 0x5EED0003)
for name, p in (("minimizer k=21 w=11", eng.params(L.MINIMIZER, 21, w=11)), ("syncmer k=31 s=11", eng.params(L.SYNCMER, 31, s=11)), ("minimizer k=15 w=5", eng.params(L.MINIMIZER, 15, w=5))):
    res, ms = eng.run_timed(u, p, 2, 5)
    uni = int(nb / 150) * 150 / min(ms) / 1e6
    res.close()
    print("== %s  uniform 150: %.1f Gbases/s" % (name, uni))
    dgs = []
    modes = [("view built with the batch", {}), ("view built per plan", {"BSK_NO_BIN_EARLY": "1"}), ("batch order", {"BSK_NO_BIN": "1"})]
    for mode, env in modes:
        for k_, v in env.items():
            os.environ[k_] = v
        t0 = time.time()
        b = eng.batch_from_arrays(data, offs)
        t_make = time.time() - t0
        prep = [eng.prepare(b, p) for _ in range(3)]
        res, ms = eng.run_timed(b, p, 2, 5)
        prep2 = min(eng.prepare(b, p) for _ in range(3))
        r = nb / min(ms) / 1e6
        rp = nb / (min(ms) + prep2) / 1e6
        dgs.append(res.digest()["checksum"])
        print("  %-28s %7.1f Gbases/s (%.3f of uniform)  with the plan's pass %7.1f (%.3f)  pass %.3f ms  batch made in %.0f ms  %s" % (
            mode, r, r / uni, rp, rp / uni, prep2, t_make * 1e3, res.plan()["kernel"]))
        res.close()
        b.close()
        for k_ in env:
            del os.environ[k_]
    assert len(set(dgs)) == 1, dgs
