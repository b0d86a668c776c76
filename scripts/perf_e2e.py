"""dev (GPU): host memory -> tuples on the host (bsk_pipeline_memory), streams x chunk size.  usage: perf_e2e.py [reads]
(BSK_PIPE_NO_COPY_LOCKS=1 in the environment: without the one-copy-per-direction locks)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bio_amd import _lib as L
from bio_amd import sketches as S

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 32_000_000
rl = 150
rng = np.random.default_rng(1)
data = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n * rl, dtype=np.uint8)].copy()
offs = (np.arange(n + 1, dtype=np.uint64) * rl)
p = S.Engine.params(L.MINIMIZER, 21, w=11)
S.Engine.pipeline_memory(data[: 2_000_000 * rl], offs[:2_000_001], p, n_streams=5, chunk_records=1 << 18, repeat=1, fetch=True)  # warm: pins
for ck in (1 << 18, 1 << 19, 1 << 20):
    for ns in (3, 5, 8):
        st = S.Engine.pipeline_memory(data, offs, p, n_streams=ns, chunk_records=ck, repeat=1, fetch=True)
        print("locks %s chunk 2^%d streams %d  %.2f Gbases/s  (h2d %.2f kern %.2f fetch %.2f pin %.2f)" % (
            "off" if os.environ.get("BSK_PIPE_NO_COPY_LOCKS") else "on", ck.bit_length() - 1, ns, st["bases"] / st["seconds"] / 1e9,
            st["h2d_pack_seconds"], st["kernel_seconds"], st["fetch_seconds"], st["pin_seconds"]), flush=True)
