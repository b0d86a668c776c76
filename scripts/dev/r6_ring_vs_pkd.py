import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ.setdefault("BSK_PY_WATCH_ENV", "1")
from bio_amd import sketches as S, _lib as L
eng = S.Engine(0)
for rl in (200, 225, 250, 275, 300):
    n = int(3e9 / rl)
    b = eng.synth(L.ALPHA_DNA, n, rl, 0x5EED0003)
    for force in ("", "1"):
        if force: os.environ["BSK_NO_RING"] = "1"
        else: os.environ.pop("BSK_NO_RING", None)
        p = eng.params(L.MINIMIZER, 21, w=11)
        res, ms = eng.run_timed(b, p, 1, 4); plan = res.plan()["kernel"]; res.close()
        walls = []
        for _ in range(5):
            t = time.time(); r2 = eng.run(b, p); walls.append(time.time() - t); r2.close()
        print(rl, plan, "kernel %.2f ms" % min(ms), "plain %s ms" % [round(w * 1e3, 2) for w in walls[1:]], flush=True)
    os.environ.pop("BSK_NO_RING", None)
    b.close()
