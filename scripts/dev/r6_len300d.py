import os, sys, time
sys.path.insert(0, "/root/repo")
from bio_amd import sketches as S, _lib as L
eng = S.Engine(0)
for rl in (200, 250, 300):
    n = int(3e9 / rl)
    b = eng.synth(L.ALPHA_DNA, n, rl, 0x5EED0003)
    for rep in range(3):
        t = time.time(); res = eng.run(b, eng.params(L.MINIMIZER, 21, w=11)); dt = time.time() - t
        print(rl, "bsk_sketch wall %.2f ms -> %.0f Gbases/s" % (dt * 1e3, n * rl / dt / 1e9), res.plan()["kernel"], res.info()["n_tuples"], flush=True)
        res.close()
    b.close()
