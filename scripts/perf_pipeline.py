"""dev (GPU): end-to-end pipeline rate against the number of streams / parser threads.  usage: perf_pipeline.py [reads]"""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bio_amd import _lib as L
from bio_amd import sketches as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
rl = 150
rng = np.random.default_rng(1)
data = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n * rl, dtype=np.uint8)].copy()
offs = (np.arange(n + 1, dtype=np.uint64) * rl)
p = S.Engine.params(L.MINIMIZER, 21, w=11)
for ns in (3, 3, 6):
    st = S.Engine.pipeline_memory(data, offs, p, n_streams=ns, chunk_records=1 << 20, repeat=2, fetch=True)
    print("memory  streams %2d  %.2f Gbases/s  (%.2f s; h2d %.2f kern %.2f fetch %.2f wait %.2f pin %.2f)" % (ns, st["bases"] / st["seconds"] / 1e9, st["seconds"], st["h2d_pack_seconds"], st["kernel_seconds"], st["fetch_seconds"], st["reader_wait_seconds"], st["pin_seconds"]), flush=True)
rec = 12 + rl + 3 + rl + 1
arr = np.empty((n, rec), np.uint8)
arr[:, 0] = ord("@"); arr[:, 1:11] = ord("r"); arr[:, 11] = 10
arr[:, 12:12 + rl] = data.reshape(n, rl)
arr[:, 12 + rl] = 10; arr[:, 13 + rl] = ord("+"); arr[:, 14 + rl] = 10; arr[:, 15 + rl:15 + 2 * rl] = ord("I"); arr[:, 15 + 2 * rl] = 10
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "r.fq")
    arr.tofile(path)
    for ns, nt in ((3, 12), (4, 11)):
        os.environ["BSK_FASTX_THREADS"] = str(nt)
        st = S.Engine.pipeline_fastx(path, p, n_streams=ns, chunk_records=1 << 20, fetch=True)
        print("file    streams %2d parsers %2d  %.2f Gbases/s  (%.2f s; reader %.2f wait %.2f h2d %.2f fetch %.2f pin %.2f)" % (ns, nt, st["bases"] / st["seconds"] / 1e9, st["seconds"], st["reader_seconds"], st["reader_wait_seconds"], st["h2d_pack_seconds"], st["fetch_seconds"], st["pin_seconds"]), flush=True)
    for ck in (1 << 18, 1 << 19, 1 << 21):
        os.environ["BSK_FASTX_THREADS"] = "10"
        st = S.Engine.pipeline_fastx(path, p, n_streams=4, chunk_records=ck, fetch=True)
        print("file    streams 4 parsers 10 chunk %d  %.2f Gbases/s" % (ck, st["bases"] / st["seconds"] / 1e9), flush=True)
