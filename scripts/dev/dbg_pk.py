import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bio_amd import sketches as S, _lib as L
from oracle import oracle
eng = S.Engine(0)
def run(lo, hi, n, seed):
    rng = random.Random(seed)
    seqs = ["".join(rng.choice("ACGT") for _ in range(rng.randint(lo, hi))) for _ in range(n)]
    b = eng.batch(seqs)
    res = eng.run(b, eng.params(L.MINIMIZER, 21, w=11))
    bad = []
    for i, s in enumerate(seqs):
        st, h, p = res.read(i)
        eh, ep, es, fl = oracle.minimizer(s, 21, 11, False, closed=True)
        if not (np.array_equal(h, eh) and np.array_equal(p & 0x7fffffff, ep)):
            bad.append((i, len(s), len(h), len(eh)))
    print(lo, hi, n, res.plan()["kernel"], "bad:", len(bad), bad[:12])
os.environ["BSK_NO_DENSE"] = "1"
run(150, 330, 200, 56)
run(150, 240, 200, 56)
run(241, 330, 200, 57)
run(150, 330, 64, 58)
run(150, 330, 128, 59)
run(300, 330, 256, 60)
run(150, 150, 200, 61)
run(150, 150, 8, 62)
run(150, 160, 72, 63)
os.environ["BSK_WAVES_PER_CU"] = "1"
run(150, 160, 200, 64)
