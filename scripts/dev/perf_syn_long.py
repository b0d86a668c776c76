"""dev (GPU): k_syncmer_pkl (round 4) against k_syncmer_fast (BSK_NO_SYN_LONG=1) over the read length, k=31 s=11 and k - s = 21..24."""
import os, sys
os.environ.setdefault("BSK_PY_WATCH_ENV", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bio_amd import sketches as S, _lib as L
eng = S.Engine(0)
cases = [(31, 11, rl) for rl in (150, 190, 200, 225, 250, 275, 300, 350)] + [(31, 10, 150), (31, 7, 150), (31, 7, 250), (35, 11, 150)]
if len(sys.argv) > 1:
    cases = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for k, s, rl in cases:
    n = int(3e9 / rl)
    b = eng.synth(L.ALPHA_DNA, n, rl, 0x5EED0003)
    out = []
    for nolong in ("0", "1"):
        os.environ.pop("BSK_NO_SYN_LONG", None)
        if nolong == "1":
            os.environ["BSK_NO_SYN_LONG"] = "1"
        res, ms = eng.run_timed(b, eng.params(L.SYNCMER, k, s=s), 1, 4)
        d = res.digest()
        out.append("%s %.0f ck%d tie%d" % (res.plan()["kernel"], n * rl / min(ms) / 1e6, d["checksum"] % 100000, d["first_window_tie"]))
        res.close()
    print("k=%d s=%d L=%d | %s" % (k, s, rl, " | ".join(out)), flush=True)
    b.close()
