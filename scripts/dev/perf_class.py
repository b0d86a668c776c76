"""dev (GPU): one class-plan batch (150-base reads + outliers), timed; for rocprofv3 --kernel-trace --stats.
usage: perf_class.py <reads> "<frac:len,frac:len...>" [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bio_amd import sketches as S, _lib as L
n = int(float(sys.argv[1])); spec = sys.argv[2]; iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rng = np.random.default_rng(1)
lens = np.full(n, 150, np.uint64)
for it in spec.split(","):
    if not it: continue
    f, ln = it.split(":")
    lens[rng.integers(0, n, int(n * float(f)))] = int(ln)
offs = np.zeros(n + 1, np.uint64); np.cumsum(lens, out=offs[1:])
data = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(offs[-1]), dtype=np.uint8)]
eng = S.Engine(0)
b = eng.batch_from_arrays(data, offs)
p = eng.params(L.MINIMIZER, 21, w=11)
res, ms = eng.run_timed(b, p, 1, iters)
print("plan:", res.plan()["kernel"])
print("kernel ms:", [round(m, 3) for m in ms], "class:", res.class_plan())
print("Gbases/s best=%.1f" % (int(offs[-1]) / min(ms) / 1e6))
print(res.digest())
