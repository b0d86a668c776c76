"""dev helper: wall-clock throughput of the tiled path on long sequences (tile build + kernel + stitch + sync)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bio_amd import sketches as S, _lib as L

total = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
nseq = int(sys.argv[2]) if len(sys.argv) > 2 else 1
eng = S.Engine(0)
rng = np.random.default_rng(1)
data = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, total, dtype=np.uint8)]
offs = np.linspace(0, total, nseq + 1).astype(np.uint64)
t = time.time(); b = eng.batch_from_arrays(data, offs); print("upload+pack s", round(time.time() - t, 3), b.info())
for name, p in (("nthash k21", eng.params(L.NTHASH, 21)), ("kmer k21", eng.params(L.KMER, 21)), ("minimizer k21 w11", eng.params(L.MINIMIZER, 21, w=11)),
                ("minimizer k31 w15", eng.params(L.MINIMIZER, 31, w=15)), ("syncmer k31 s11", eng.params(L.SYNCMER, 31, s=11))):
    ts = []
    for _ in range(3):
        t = time.time(); res = eng.run(b, p); ts.append(time.time() - t); nt = res.info()["n_tuples"]; res.close()
    res, ms = eng.run_timed(b, p, 1, 3); res.close()
    print(f"{name:20s} tile-kernel ms {[round(m,3) for m in ms]} -> {total/min(ms)/1e6:.0f} Gbases/s in the kernel")
    print(f"{name:20s} wall ms {[round(x*1e3,1) for x in ts]} -> {total/min(ts)/1e9:.1f} Gbases/s, tuples {nt}")
