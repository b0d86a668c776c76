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
for ns, rep in ((3, 8), (3, 8), (5, 8), (5, 8), (8, 8)):
  for fetch in (True,):
    st = S.Engine.pipeline_memory(data, offs, p, n_streams=ns, chunk_records=1 << 18, repeat=rep, fetch=fetch)
    print("memory  rep %d streams %2d fetch %d %.2f Gbases/s  (%.2f s; h2d %.2f kern %.2f fetch %.2f wait %.2f pin %.2f)" % (rep, ns, fetch, st["bases"] / st["seconds"] / 1e9, st["seconds"], st["h2d_pack_seconds"], st["kernel_seconds"], st["fetch_seconds"], st["reader_wait_seconds"], st["pin_seconds"]), flush=True)
