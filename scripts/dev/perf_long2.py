import sys, time, os
import numpy as np
sys.path.insert(0, "/root/repo")
from bio_amd import sketches as S, _lib as L
total = int(float(os.environ.get('TOTAL', '2e9'))); nseq = int(os.environ.get('NSEQ', '400'))
eng = S.Engine(0)
rng = np.random.default_rng(1)
data = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, total, dtype=np.uint8)]
offs = np.linspace(0, total, nseq + 1).astype(np.uint64)
b = eng.batch_from_arrays(data, offs)
only = os.environ.get("ONLY", "")
for name, p in (("minimizer k21 w11", eng.params(L.MINIMIZER, 21, w=11)), ("syncmer k31 s11", eng.params(L.SYNCMER, 31, s=11))):
    if only and only not in name: continue
    ts = []
    for _ in range(4):
        t = time.time(); res = eng.run(b, p); ts.append(time.time() - t); nt = res.info()["n_tuples"]; pl = res.plan()["kernel"]; dg = 0 if os.environ.get("NODIGEST") else res.digest()["checksum"]; res.close()
    print(f"{name:20s} wall ms {[round(x*1e3,1) for x in ts]} -> {total/min(ts)/1e9:.1f} Gbases/s, tuples {nt} {pl} {dg}")
