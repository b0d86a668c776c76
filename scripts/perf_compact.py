"""dev (GPU): bsk_result_compact (dense CSR copy of a result left on the device) timed behind the sketch it follows.
usage: perf_compact.py [n] [kind] [k] [x] [len]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bio_amd import sketches as S, _lib as L

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "min"
k = int(sys.argv[3]) if len(sys.argv) > 3 else 21
x = int(sys.argv[4]) if len(sys.argv) > 4 else 11
rlen = int(sys.argv[5]) if len(sys.argv) > 5 else 0
eng = S.Engine(0)
prot = kind.startswith("p")
LL = rlen or (300 if prot else 150)
b = eng.synth(L.ALPHA_PROTEIN if prot else L.ALPHA_DNA, n, LL, 0x5EED0003)
p = {"min": eng.params(L.MINIMIZER, k, w=x), "syn": eng.params(L.SYNCMER, k, s=x), "nt": eng.params(L.NTHASH, k), "pmin": eng.params(L.PROT_MINIMIZER, k, w=x)}[kind]
res = eng.run(b, p)
res, ms = eng.run_timed(b, p, 1, 3, reuse=res)
print("kernel", res.plan()["kernel"], "ms", [round(m, 3) for m in ms])
ts = []
for _ in range(4):
    t = time.perf_counter()
    po, ph, pp, nt = res.compact()
    ts.append((time.perf_counter() - t) * 1e3)
bytes_moved = nt * 24 + n * 24  # tuples read + written, reference words read, offsets written + read
print("compact ms (wall, incl. its two synchronisations):", [round(t, 3) for t in ts], "tuples", nt)
best = min(ts[1:])
print(f"compact: {best:.3f} ms = {bytes_moved / best / 1e6:.0f} GB/s of {bytes_moved / 1e9:.2f} GB moved; sketch + compact {n * LL / (min(ms) + best) / 1e6:.1f} Gbases/s (sketch alone {n * LL / min(ms) / 1e6:.1f})")
print(res.digest())
