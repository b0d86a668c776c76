"""dev (GPU): k_minimizer_pkd against the kernels it would replace, over the read length: the shipped plan, BSK_NO_RING=1 (pkd from ~160 bases),
BSK_NO_PKD=1 (ring / dense as before), digests compared.  usage: python scripts/dev/perf_pkd.py [bases] [w] [k]"""
import os, sys
os.environ.setdefault("BSK_PY_WATCH_ENV", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bio_amd import sketches as S, _lib as L

BASES = float(sys.argv[1]) if len(sys.argv) > 1 else 3e9
W = int(sys.argv[2]) if len(sys.argv) > 2 else 11
K = int(sys.argv[3]) if len(sys.argv) > 3 else 21
eng = S.Engine(0)
p = eng.params(L.MINIMIZER, K, w=W)
for rl in (150, 180, 200, 250, 300, 350, 400, 600, 1000, 3000):
    n = int(BASES / rl)
    b = eng.synth(L.ALPHA_DNA, n, rl, 0x5EED0003)
    row, dg = [], []
    for mode, env in (("shipped", {}), ("no ring", {"BSK_NO_RING": "1"}), ("no pkd", {"BSK_NO_PKD": "1"}), ("neither", {"BSK_NO_PKD": "1", "BSK_NO_RING": "1"})):
        for k_, v in env.items():
            os.environ[k_] = v
        res, ms = eng.run_timed(b, p, 2, 5)
        row.append("%s %7.1f %s" % (mode, n * rl / min(ms) / 1e6, res.plan()["kernel"].replace("k_minimizer_", "")))
        dg.append(res.digest()["checksum"])
        res.close()
        for k_ in env:
            del os.environ[k_]
    print("%5d bp | " % rl + " | ".join(row) + (" | digests equal" if len(set(dg)) == 1 else " | DIGESTS DIFFER %s" % dg), flush=True)
    b.close()
