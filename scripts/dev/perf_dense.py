"""dev (GPU): k_minimizer_dense over read lengths / windows the other kernels leave to it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bio_amd import sketches as S, _lib as L
eng = S.Engine(0)
for w, rl in ((11, 400), (11, 600), (11, 1000), (11, 2000), (5, 150), (5, 250), (3, 150), (15, 400)):
    n = int(3e9 / rl)
    b = eng.synth(L.ALPHA_DNA, n, rl, 0x5EED0003)
    res, ms = eng.run_timed(b, eng.params(L.MINIMIZER, 21, w=w), 1, 4)
    print("w=%d L=%d %s %.0f Gb/s %d" % (w, rl, res.plan()["kernel"], n * rl / min(ms) / 1e6, res.digest()["checksum"] % 100000), flush=True)
    res.close(); b.close()
