import os, sys
os.environ.setdefault("BSK_PY_WATCH_ENV", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bio_amd import sketches as S, _lib as L
eng = S.Engine(0)
p = eng.params(L.MINIMIZER, 21, w=11)
for rl in (300, 350):
    n = int(3e9 / rl)
    b = eng.synth(L.ALPHA_DNA, n, rl, 0x5EED0003)
    keep = []
    for i in range(8):
        res, ms = eng.run_timed(b, p, 1, 4)
        print(rl, i, "%.1f" % (n * rl / min(ms) / 1e6), [round(m, 3) for m in ms], res.plan()["kernel"], res.info().get("capacity"), flush=True)
        if i % 2: keep.append(res)
        else: res.close()
    for r in keep: r.close()
    b.close()
