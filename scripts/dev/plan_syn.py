import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bio_amd import sketches as S, _lib as L
eng = S.Engine(0)
b = eng.synth(L.ALPHA_DNA, 10**6, 150, 3)
r = eng.run(b, eng.params(L.SYNCMER, 31, s=11))
print(r.plan())
