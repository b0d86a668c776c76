"""dev (GPU): one seed of tests/test_gpu_fuzz.py in a process of its own, with what it drew.  usage: fuzz_one.py seed"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("BSK_PY_WATCH_ENV", "1")
import test_gpu_fuzz as F
from bio_amd import sketches as S, _lib as L
from oracle import oracle
seed = int(sys.argv[1])
rng = random.Random(seed)
kind = rng.choice([L.MINIMIZER, L.MINIMIZER, L.SYNCMER, L.NTHASH, L.KMER, L.SIMHASH, L.PROT_HASH, L.PROT_MINIMIZER])
print("seed", seed, "kind", kind, "env", F.env_switches(rng), flush=True)
eng = S.Engine(0)
F.run_case(eng, oracle, seed)
eng.sync()
print("seed", seed, "ok", flush=True)
