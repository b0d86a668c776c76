"""dev (GPU): syncmer k=31 s=11 against the read length, and k_syncmer_pk with less slack in its staging columns (BSK_SYN_MARGIN = 64 + rows)."""
import os, sys
os.environ.setdefault("BSK_PY_WATCH_ENV", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bio_amd import sketches as S, _lib as L
eng = S.Engine(0)
for rl in (150, 165, 180, 190, 200, 250):
    n = int(3e9 / rl)
    b = eng.synth(L.ALPHA_DNA, n, rl, 0x5EED0003)
    out = []
    for m in (70, 66, 64, 58):
        os.environ["BSK_SYN_MARGIN"] = str(m)
        res, ms = eng.run_timed(b, eng.params(L.SYNCMER, 31, s=11), 1, 4)
        out.append("%d:%s %.0f %d" % (m - 64, res.plan()["kernel"].split("<")[0][-4:], n * rl / min(ms) / 1e6, res.digest()["checksum"] % 100000))
        res.close()
    print("L=%d | %s" % (rl, " | ".join(out)), flush=True)
    b.close()
