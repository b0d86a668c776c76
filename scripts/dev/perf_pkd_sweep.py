"""dev (GPU): where k_minimizer_pkd overtakes k_minimizer_ring, per window size: expected tuples per read = windows x 2 / (w + 1).
usage: python scripts/dev/perf_pkd_sweep.py [bases]"""
import os, sys
os.environ.setdefault("BSK_PY_WATCH_ENV", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bio_amd import sketches as S, _lib as L

BASES = float(sys.argv[1]) if len(sys.argv) > 1 else 2e9
eng = S.Engine(0)
for w, k in [(int(x.split(":")[0]), int(x.split(":")[1])) for x in os.environ.get("WK", "5:15,8:21,11:21,13:21").split(",")]:
    p = eng.params(L.MINIMIZER, k, w=w)
    for rl in [int(x) for x in os.environ.get("RLS", "175,200,225,250,275,300,325,350,400").split(",")]:
        n = int(BASES / rl)
        b = eng.synth(L.ALPHA_DNA, n, rl, 0x5EED0003)
        best, allms = {}, {"ring": [], "pkd": []}
        for rep in range(3):
            for mode, env in (("ring", {"BSK_RING_MAX": "100000"}), ("pkd", {"BSK_NO_RING": "1"})):
                for k_, v in env.items():
                    os.environ[k_] = v
                res, ms = eng.run_timed(b, p, 1, 4)
                if rep:  # (the first round warms the board up: a launch right after an idle stretch runs above the power cap's steady clock)
                    allms[mode] += list(ms)
                    m = sorted(allms[mode])
                    best[mode] = n * rl / m[len(m) // 2] / 1e6
                best[mode + "_plan"] = res.plan()["kernel"].replace("k_minimizer_", "")
                res.close()
                for k_ in env:
                    del os.environ[k_]
        exp = (rl - k - w + 2) * 2.0 / (w + 1)
        print("w=%2d k=%2d %4d bp  expected tuples %5.1f  ring %7.1f (%s)  pkd %7.1f (%s)  %s" % (w, k, rl, exp, best["ring"], best["ring_plan"], best["pkd"], best["pkd_plan"],
                                                                                          "pkd" if best["pkd"] > best["ring"] else "ring"), flush=True)
        b.close()
