"""dev (GPU): k_minimizer_ring (BSK_RING=1) against the planner's other choice over window sizes and read lengths."""
import sys, os
os.environ.setdefault("BSK_PY_WATCH_ENV", "1")  # this script flips BSK_* switches between runs
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bio_amd import sketches as S, _lib as L
eng = S.Engine(0)
ws = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [11]
ls = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [150, 250]
for w in ws:
    for rl in ls:
        n = int(1.5e9 / rl)
        b = eng.synth(L.ALPHA_DNA, n, rl, 0x5EED0003)
        out = []
        for ring in (1, 0):
            os.environ.pop("BSK_RING", None), os.environ.pop("BSK_NO_RING", None)
            os.environ["BSK_RING" if ring else "BSK_NO_RING"] = "1"
            p = eng.params(L.MINIMIZER, 21, w=w)
            res, ms = eng.run_timed(b, p, 1, 4)
            out.append("%s %.0f Gb/s %s" % (res.plan()["kernel"], n * rl / min(ms) / 1e6, res.digest()["checksum"] % 100000))
            res.close()
        print("w=%d L=%d | %s | %s" % (w, rl, out[0], out[1]), flush=True)
        b.close()
