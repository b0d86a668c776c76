"""dev helper: run the differential fuzz of tests/test_gpu_fuzz.py over many more seeds.  usage: fuzz_campaign.py first count"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_fuzz as F
from bio_amd import sketches as S
from oracle import oracle
first, count = int(sys.argv[1]), int(sys.argv[2])
eng = S.Engine(0)
bad = 0
for seed in range(first, first + count):
    if os.environ.get("FUZZ_TRACE"):
        print("seed", seed, flush=True)
    try:
        F.run_case(eng, oracle, seed)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("SEED", seed, "FAILED:", repr(e)[:400])
        if bad >= 5:
            break
print("done", count, "cases,", bad, "failures")
