"""dev helper: time bsk_result_sets (device-side sorted distinct hash sets) on a synthetic batch's sketch.
usage: perf_sets.py [n_reads] [kind min|syn] [k] [x] [read_len]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bio_amd import _lib as L
from bio_amd import sketches as S

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "min"
k = int(sys.argv[3]) if len(sys.argv) > 3 else 21
x = int(sys.argv[4]) if len(sys.argv) > 4 else 11
rlen = int(sys.argv[5]) if len(sys.argv) > 5 else 150
eng = S.Engine(0)
b = eng.synth(L.ALPHA_DNA, n, rlen, 0x5EED0003)
p = eng.params(L.MINIMIZER, k, w=x) if kind == "min" else eng.params(L.SYNCMER, k, s=x)
res = eng.run(b, p)
T = res.info()["n_tuples"]
for scope, name in ((L.SETS_PER_SEQUENCE, "per sequence"), (L.SETS_WHOLE_BATCH, "whole batch")):
    for scale in (1, 10):
        best = 1e9
        for it in range(3):
            h = C.c_void_p()
            eng.lib.bsk_ctx_sync(eng.ctx)
            t = time.perf_counter()
            eng._chk(eng.lib.bsk_result_sets(eng.ctx, res.h, scope, scale, C.byref(h)))
            dt = time.perf_counter() - t
            ns, nv = C.c_uint64(), C.c_uint64()
            eng.lib.bsk_sets_info(h, C.byref(ns), C.byref(nv))
            eng.lib.bsk_sets_release(h)
            best = min(best, dt)
        print(f"sets {name:12s} scale={scale:2d}: {best*1e3:8.2f} ms  {T/best/1e9:6.2f} G tuples/s in, {nv.value} values out in {ns.value} sets "
              f"({n*rlen/best/1e9:.1f} Gbases/s of the sketched reads)")
