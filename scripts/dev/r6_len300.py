import os, sys
sys.path.insert(0, "/root/repo")
from bio_amd import sketches as S, _lib as L
eng = S.Engine(0)
for rl in (250, 300, 350):
    n = int(3e9 / rl)
    b = eng.synth(L.ALPHA_DNA, n, rl, 0x5EED0003)
    for rep in range(2):
        res, ms = eng.run_timed(b, eng.params(L.MINIMIZER, 21, w=11), 2, 5)
        print(rl, "%.1f Gbases/s" % (n * rl / min(ms) / 1e6), [round(x, 2) for x in ms], res.plan()["kernel"], res.info()["n_tuples"], flush=True)
        res.close()
    b.close()
