"""dev (GPU): reads between the packed kernels' limits and the tiling threshold (4 096 bases): syncmer k=31 s=11 and minimizer k=21 w=11,
as planned | as planned before round 4's long syncmer plan (BSK_NO_SYN_LONG=1: k_syncmer_fast up to 4 096 bases, tiles of 112 positions beyond) |
with tiles forced from 384 bases (BSK_TILE_MIN=384).  Wall time of run() (tiles pay a table build + a stitch pass)."""
import os, sys, time
os.environ.setdefault("BSK_PY_WATCH_ENV", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bio_amd import sketches as S, _lib as L
eng = S.Engine(0)
for kind, k, x in ((L.SYNCMER, 31, 11), (L.MINIMIZER, 21, 11)):
    for rl in (400, 450, 500, 700, 1000, 2000, 4000, 5000, 100000):
        n = int(2e9 / rl)
        b = eng.synth(L.ALPHA_DNA, n, rl, 0x5EED0003)
        prm = eng.params(kind, k, s=x) if kind == L.SYNCMER else eng.params(kind, k, w=x)
        out = []
        for tm in (None, "nolong", "384"):
            os.environ.pop("BSK_TILE_MIN", None)
            os.environ.pop("BSK_NO_SYN_LONG", None)
            if tm == "nolong":
                os.environ["BSK_NO_SYN_LONG"] = "1"
            elif tm:
                os.environ["BSK_TILE_MIN"] = tm
            eng.reload_options()
            res = eng.run(b, prm); res.close()
            ts = []
            for _ in range(3):
                t = time.time(); res = eng.run(b, prm); ts.append(time.time() - t)
                d = res.digest(); pl = res.plan()["kernel"]; res.close()
            out.append("%s %.0f ck%d" % (pl, n * rl / min(ts) / 1e9, d["checksum"] % 100000))
        print("%s L=%d | %s" % ("syn" if kind == L.SYNCMER else "min", rl, " | ".join(out)), flush=True)
        b.close()
