import sys, time, os
import numpy as np
sys.path.insert(0, "/root/repo")
from bio_amd import sketches as S, _lib as L
n = int(float(os.environ.get("N", "3e7"))); ln = 150
eng = S.Engine(0)
rng = np.random.default_rng(3)
data = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n * ln, dtype=np.uint8)].copy()
bad = rng.integers(0, n, n // 100)
data[bad * ln + 75] = ord("N")
offs = (np.arange(n + 1, dtype=np.uint64) * ln)
for rep in range(3):
    t = time.time(); b = eng.batch_from_arrays(data, offs); t1 = time.time() - t
    print(f"batch_from_ascii {n} reads, 1 % with an N: {t1*1e3:.1f} ms")
    if rep < 2: b.close()
data2 = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n * ln, dtype=np.uint8)]
b2 = eng.batch_from_arrays(data2, offs)
for rep in range(3):
    t = time.time(); pb = b2.translate(1, 1); t1 = time.time() - t
    print(f"translate {n} reads x {ln}: {t1*1e3:.2f} ms -> {n*ln/t1/1e9:.0f} Gbases/s")
    pb.close()
