"""dev helper: time one kind on a synthetic batch; prints Gbases/s + digest."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bio_amd import sketches as S, _lib as L

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "min"
k = int(sys.argv[3]) if len(sys.argv) > 3 else 21
x = int(sys.argv[4]) if len(sys.argv) > 4 else 11
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
rlen = int(sys.argv[6]) if len(sys.argv) > 6 else 0  # read length override
eng = S.Engine(0)
t = time.time()
prot = kind.startswith("p")
b = eng.synth(L.ALPHA_PROTEIN, n, rlen or 300, 0x5EED0005) if prot else eng.synth(L.ALPHA_DNA, n, rlen or 150, 0x5EED0003)
print("synth", time.time() - t, b.info())
p = {"min": eng.params(L.MINIMIZER, k, w=x), "nt": eng.params(L.NTHASH, k), "syn": eng.params(L.SYNCMER, k, s=x), "pmin": eng.params(L.PROT_MINIMIZER, k, w=x), "phash": eng.params(L.PROT_HASH, k),
     "kmer": eng.params(L.KMER, k), "sim": eng.params(L.SIMHASH, k, m=5, scale=5)}[kind]
t = time.time()
res, ms = eng.run_timed(b, p, 1, iters)
wall = time.time() - t
inf = res.info()
best = min(ms)
print(f"kind={kind} k={k} x={x} n={n} tuples={inf['n_tuples']} per_read={inf['n_tuples']/n:.2f}")
print("kernel ms:", [round(m, 3) for m in ms], "wall", round(wall, 3))
LL = rlen or (300 if prot else 150)
print(f"Gbases/s best={n*LL/best/1e6:.1f} avg={n*LL/(sum(ms)/len(ms))/1e6:.1f}")
print("plan:", res.plan()["kernel"])
print(res.digest())
