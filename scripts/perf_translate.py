"""dev helper: time bsk_batch_translate (wall clock incl. allocation + sync) and the DNA-fed protein sketch."""
import sys, time, os
os.environ.setdefault("BSK_PY_WATCH_ENV", "1")  # this script flips BSK_* switches between runs
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bio_amd import sketches as S, _lib as L

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 5_000_000
length = int(sys.argv[2]) if len(sys.argv) > 2 else 900
eng = S.Engine(0)
b = eng.synth(L.ALPHA_DNA, n, length, 0x5EED0007)
for frame in (1, -2):
    ts = []
    for _ in range(4):
        t = time.time()
        tb = b.translate(1, frame)
        ts.append(time.time() - t)
        info = tb.info()
        tb.close()
    print(f"translate frame {frame}: n={n} L={length} residues={info['n_bases']} wall ms {[round(x*1e3,2) for x in ts]} -> {n*length/min(ts)/1e9:.1f} Gbases/s")
p = eng.params(L.PROT_MINIMIZER, 9, w=5)
t = time.time(); res = eng.run(b, p); eng.sync() if hasattr(eng, "sync") else None; t1 = time.time() - t
t = time.time(); res = eng.run(b, p, res); t2 = time.time() - t
print(f"DNA-fed protein minimizer (translate + sketch) wall ms first {t1*1e3:.1f} repeat {t2*1e3:.1f}; tuples {res.info()['n_tuples']}")
tb = b.translate(1, 1)
res2, ms = eng.run_timed(tb, p, 1, 3)
print("protein-batch sketch kernel ms", [round(m, 3) for m in ms], "tuples", res2.info()["n_tuples"])
# fused (the kernel translates where it fetches residues) against the two-step path, same digest
res_f, ms_f = eng.run_timed(b, p, 1, 3)
os.environ["BSK_NO_FUSED_TRANSLATE"] = "1"
t = time.time(); res_u = eng.run(b, p); t3 = time.time() - t
t = time.time(); res_u = eng.run(b, p, res_u); t4 = time.time() - t
del os.environ["BSK_NO_FUSED_TRANSLATE"]
print("fused DNA-fed kernel ms", [round(m, 3) for m in ms_f], res_f.plan()["kernel"], "| two-step call wall ms", round(t4 * 1e3, 2), res_u.plan()["kernel"])
assert res_f.digest() == res_u.digest() == res2.digest(), (res_f.digest(), res_u.digest(), res2.digest())
print("digests equal:", res_f.digest())
