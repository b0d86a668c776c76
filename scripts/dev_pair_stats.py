"""dev: per-read tuple counts of a synthetic batch and how often a paired staging column of R rows would fill up.
usage: dev_pair_stats.py [min|syn] k x [n] [len]"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bio_amd import sketches as S, _lib as L
kind = sys.argv[1] if len(sys.argv) > 1 else "min"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 21
x = int(sys.argv[3]) if len(sys.argv) > 3 else 11
n = int(float(sys.argv[4])) if len(sys.argv) > 4 else 6400000
ln = int(sys.argv[5]) if len(sys.argv) > 5 else 150
eng = S.Engine(0)
b = eng.synth(L.ALPHA_DNA, n, ln, 0x5EED0003)
p = eng.params(L.MINIMIZER, k, w=x) if kind == "min" else eng.params(L.SYNCMER, k, s=x)
res = eng.run(b, p)
offs, st, h, pp = res.fetch(0, n)
c = np.diff(offs).astype(np.int64)
print(kind, k, x, "mean", c.mean(), "sd", c.std(), "max", c.max())
u = c.reshape(-1, 64)
pair = u[:, :32] + u[:, 32:]
print("pair mean", pair.mean(), "sd", pair.std(), "max", pair.max())
lo = int(pair.mean())
for R in range(lo + 4, lo + 22, 2):
    print(R, "cols", (pair >= R).mean(), "units", (pair >= R).any(axis=1).mean())
