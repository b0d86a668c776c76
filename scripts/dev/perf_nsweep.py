"""dev (GPU): kernel time of the planned minimizer kernel against the batch size -- the fixed cost of a launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bio_amd import sketches as S, _lib as L
eng = S.Engine(0)
rl = int(sys.argv[1]) if len(sys.argv) > 1 else 150
for n in [1 << 16, 1 << 18, 1 << 20, 1 << 22, 10_000_000, 30_000_000]:
    b = eng.synth(L.ALPHA_DNA, n, rl, 0x5EED0003)
    p = eng.params(L.MINIMIZER, 21, w=11)
    res, ms = eng.run_timed(b, p, 2, 8)
    print("n=%9d %-28s min %.4f ms  median %.4f ms  %.0f Gb/s" % (n, res.plan()["kernel"], min(ms), sorted(ms)[len(ms) // 2], n * rl / min(ms) / 1e6), flush=True)
    res.close()
    b.close()
