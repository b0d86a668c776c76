import os, sys
os.environ.setdefault("BSK_PY_WATCH_ENV", "1")
os.environ["BSK_NO_TILES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bio_amd import sketches as S, _lib as L
eng = S.Engine(0)
p = eng.params(L.MINIMIZER, 21, w=11)
for rl in (500, 1000, 2000, 3000, 5000, 10000, 20000):
    n = int(1.2e10 / rl)
    b = eng.synth(L.ALPHA_DNA, n, rl, 0x5EED0003)
    row = []
    for mode, env in (("pkd", {}), ("dense", {"BSK_NO_PKD": "1"})):
        for k_, v in env.items(): os.environ[k_] = v
        res, ms = eng.run_timed(b, p, 1, 3)
        row.append("%s %7.1f %s" % (mode, n * rl / min(ms) / 1e6, res.plan()["kernel"]))
        res.close()
        for k_ in env: del os.environ[k_]
    print("%6d bp | " % rl + " | ".join(row), flush=True)
    b.close()
