import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from bio_amd import sketches as S, _lib as L
eng = S.Engine(0)
n = 6400000
b = eng.synth(L.ALPHA_DNA, n, 150, 0x5EED0003)
res = eng.run(b, eng.params(L.MINIMIZER, 21, w=11))
offs, st, h, p = res.fetch(0, n)
c = np.diff(offs).astype(np.int64)
print("mean", c.mean(), "sd", c.std(), "max", c.max())
u = c.reshape(-1, 64)
pair = u[:, :32] + u[:, 32:]
print("pair mean", pair.mean(), "sd", pair.std(), "max", pair.max())
for R in (54, 55, 56, 57, 58, 59, 60):
    print(R, "cols", (pair >= R).mean(), "units", (pair >= R).any(axis=1).mean())
