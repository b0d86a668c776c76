import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from bio_amd import _lib as L
from bio_amd import sketches as S
n = 16_000_000; rl = 150
rng = np.random.default_rng(1)
data = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n * rl, dtype=np.uint8)].copy()
offs = (np.arange(n + 1, dtype=np.uint64) * rl)
p = S.Engine.params(L.MINIMIZER, 21, w=11)
for ns, rep, ck in ((5, 8, 18), (5, 8, 18), (5, 8, 19), (5, 8, 19), (5, 8, 20), (5, 8, 20), (3, 8, 20), (3, 8, 20), (5, 8, 17), (5, 8, 17)):
  for fetch in (True,):
    st = S.Engine.pipeline_memory(data, offs, p, n_streams=ns, chunk_records=1 << ck, repeat=rep, fetch=fetch)
    print("memory  chunk 2^%d rep %d streams %2d fetch %d %.2f Gbases/s  (%.2f s; h2d %.2f kern %.2f fetch %.2f wait %.2f pin %.2f)" % (ck, rep, ns, fetch, st["bases"] / st["seconds"] / 1e9, st["seconds"], st["h2d_pack_seconds"], st["kernel_seconds"], st["fetch_seconds"], st["reader_wait_seconds"], st["pin_seconds"]), flush=True)
