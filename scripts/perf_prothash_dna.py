import sys, time, os
sys.path.insert(0, "/root/repo")
from bio_amd import sketches as S, _lib as L
eng = S.Engine(0)
n, length = 5_000_000, 900
b = eng.synth(L.ALPHA_DNA, n, length, 0x5EED0007)
p = eng.params(L.PROT_HASH, 9, codon_table=1, frame=1)
res = eng.run(b, p)
ts=[]
for _ in range(4):
    t=time.time(); res = eng.run(b, p, res); ts.append(time.time()-t)
print("DNA-fed prot hash wall ms", [round(x*1e3,2) for x in ts], res.plan())
for frame in (1, -2):
    pf = eng.params(L.PROT_HASH, 9, codon_table=1, frame=frame)
    rf, msf = eng.run_timed(b, pf, 1, 5)
    print("DNA-fed (fused) kernel ms frame", frame, [round(m, 3) for m in msf], rf.plan()["kernel"], "waves/CU", rf.plan()["waves_per_cu"])
tb = b.translate(1, 1)
res2, ms = eng.run_timed(tb, p, 1, 3)
print("protein-fed kernel ms", [round(m,3) for m in ms])
ts=[]
for _ in range(3):
    t=time.time(); x = b.translate(1,1); ts.append(time.time()-t); x.close()
print("translate wall ms", [round(x*1e3,2) for x in ts])
