import os, sys
sys.path.insert(0, "/root/repo")
os.environ.setdefault("BSK_PY_WATCH_ENV", "1")
from bio_amd import sketches as S, _lib as L
eng = S.Engine(0)
for rl in (200, 250, 300):
    n = int(3e9 / rl)
    b = eng.synth(L.ALPHA_DNA, n, rl, 0x5EED0003)
    for sel in (0, 30, 34, 40):
        if sel: os.environ["BSK_RING_SEL10"] = str(sel)
        else: os.environ.pop("BSK_RING_SEL10", None)
        for rep in range(2):
            res, ms = eng.run_timed(b, eng.params(L.MINIMIZER, 21, w=11), 1, 4)
            print(rl, "sel", sel, "%.1f Gbases/s" % (n * rl / min(ms) / 1e6), [round(x, 2) for x in ms], res.plan()["kernel"], res.info()["n_tuples"], flush=True)
            res.close()
    b.close()
