import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bio_amd import sketches as S, _lib as L
from oracle import oracle
eng = S.Engine(0)
os.environ["BSK_NO_DENSE"] = "1"
for trial in range(4):
    rng = random.Random(61)
    n = 200
    seqs = ["".join(rng.choice("ACGT") for _ in range(150)) for _ in range(n)]
    b = eng.batch(seqs)
    res = eng.run(b, eng.params(L.MINIMIZER, 21, w=11))
    exp = [oracle.minimizer(s, 21, 11, False, closed=True)[0] for s in seqs]
    out = []
    for i in range(n):
        st, h, p = res.read(i)
        if not np.array_equal(h, exp[i]):
            m = [j for j in range(n) if np.array_equal(h, exp[j])]
            # partial match: which read's hashes are these
            pm = [j for j in range(n) if len(set(h.tolist()) & set(exp[j].tolist())) > 5]
            out.append((i, m, pm))
    print("trial", trial, "bad", len(out), out[:6], out[-3:])
