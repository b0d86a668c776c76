import os, sys, time
sys.path.insert(0, "/root/repo")
from bio_amd import sketches as S, _lib as L
eng = S.Engine(0)
cases = [("minimizer k21 w11", L.MINIMIZER, dict(k=21, w=11), 150, int(1e8)), ("syncmer k31 s11", L.SYNCMER, dict(k=31, s=11), 150, int(1.25e8)),
         ("minimizer 400", L.MINIMIZER, dict(k=21, w=11), 400, int(3.75e7)), ("syncmer 250", L.SYNCMER, dict(k=31, s=11), 250, int(6e7)),
         ("nthash k21", L.NTHASH, dict(k=21), 150, int(1e7)), ("kmer two strands", L.KMER, dict(k=21, canonical=False), 150, int(1e7))]
for name, kind, par, rl, n in cases:
    b = eng.synth(L.ALPHA_DNA, n, rl, 0x5EED0003)
    p = eng.params(kind, **par)
    res, ms = eng.run_timed(b, p, 1, 3); res.close()
    walls = []
    for rep in range(4):
        t = time.time(); res = eng.run(b, p); walls.append(time.time() - t); res.close()
    print("%-22s %d x %d: kernel %.2f ms, plain bsk_sketch calls %s ms" % (name, n, rl, min(ms), [round(w * 1e3, 2) for w in walls]), flush=True)
    b.close()
