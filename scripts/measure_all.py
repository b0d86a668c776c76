#!/usr/bin/env python3
"""Measures every iterator kind of the path on the GPU (device-resident synthetic input, HIP-event kernel time)
next to the CPU oracle on a bounded sample, and prints/writes one JSON table.

    python scripts/measure_all.py [out.json]

Algorithmic bytes follow SURVEY.md 8d: input = ceil(L/4)+8 per DNA read (L+8 per protein sequence);
output = 12 B per (hash,pos) tuple + 8 B per read for the sketches, 8 B per value for the every-position kinds.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from bio_amd import _lib as L  # noqa: E402
from bio_amd import sketches as S  # noqa: E402
from oracle import oracle as O  # noqa: E402

eng = S.Engine(0)
cores = len(os.sched_getaffinity(0))
CASES = [
    # name, reference row, alphabet, n, len, params, oracle kind, (k, x)
    ("A1 KmerIterator k=31 canonical", L.ALPHA_DNA, 20_000_000, 150, eng.params(L.KMER, 31), 1, (31, 0)),
    ("A2 HashIterator k=21 canonical (BASELINE configs[1])", L.ALPHA_DNA, 10_000_000, 150, eng.params(L.NTHASH, 21), 2, (21, 0)),
    ("A3 SimHashIterator k=31 m=5 scale=5", L.ALPHA_DNA, 5_000_000, 150, eng.params(L.SIMHASH, 31, m=5, scale=5), 3, (31, 0)),
    ("A5 MinimizerSketch k=21 w=11 (BASELINE configs[2])", L.ALPHA_DNA, 100_000_000, 150, eng.params(L.MINIMIZER, 21, w=11), 4, (21, 11)),
    ("A5 MinimizerSketch k=31 w=15 (reference README)", L.ALPHA_DNA, 20_000_000, 150, eng.params(L.MINIMIZER, 31, w=15), 4, (31, 15)),
    ("A6 SyncmerSketch k=31 s=11 (BASELINE configs[3], one GPU)", L.ALPHA_DNA, 50_000_000, 150, eng.params(L.SYNCMER, 31, s=11), 5, (31, 11)),
    ("A6 SyncmerSketch k=31 s=16 (reference README)", L.ALPHA_DNA, 20_000_000, 150, eng.params(L.SYNCMER, 31, s=16), 5, (31, 16)),
    ("A7 ProteinIterator k=9", L.ALPHA_PROTEIN, 5_000_000, 300, eng.params(L.PROT_HASH, 9), 6, (9, 0)),
    ("A8 ProteinMinimizerSketch k=9 w=5 (BASELINE configs[4])", L.ALPHA_PROTEIN, 50_000_000, 300, eng.params(L.PROT_MINIMIZER, 9, w=5), 7, (9, 5)),
]
rows = []
rng = np.random.default_rng(7)
for name, alpha, n, ln, p, okind, (k, x) in CASES:
    b = eng.synth(alpha, n, ln, 0x5EED0000 + okind)
    res = eng.run(b, p)
    res, ms = eng.run_timed(b, p, 1, 5, reuse=res)
    inf = res.info()
    T = inf["n_tuples"]
    a_in = n * (((ln + 3) // 4 + 8) if alpha == L.ALPHA_DNA else (ln + 8))
    a_out = (12 * T + 8 * n) if inf["has_pos"] else 8 * T
    kms = sum(ms) / len(ms)
    # CPU oracle on a bounded sample of the same generator
    ns = 100_000 if okind in (3,) else 400_000
    sb = eng.synth(alpha, ns, ln, 0x5EED0000 + okind)
    data, offs = sb.fetch_ascii(0, ns)
    t = time.perf_counter()
    nt, ck = O.batch_run(okind, data, offs, k, x, threads=cores)
    cpu = ns * ln / (time.perf_counter() - t) / 1e9
    d = eng.run(sb, p).digest()
    assert (d["n_tuples"], d["checksum"]) == (nt, ck), name
    row = {"case": name, "units": n * ln, "reads": n, "tuples": T, "kernel_ms": round(kms, 4),
           "G_units_per_s": round(n * ln / kms / 1e6, 1), "algorithmic_GB": round((a_in + a_out) / 1e9, 3),
           "achieved_GBps": round((a_in + a_out) / kms / 1e6, 1), "hbm_frac_of_8TBps": round((a_in + a_out) / kms / 1e6 / 8000, 4),
           "cpu_oracle_G_units_per_s": round(cpu, 4), "cpu_threads": cores, "parity_sample": f"{ns} reads: digest == oracle"}
    rows.append(row)
    print(json.dumps(row), flush=True)
    del res, b
# f4: sketch SETS on the device (bsk_result_sets): sorted distinct hashes per sequence / per batch, plain and FracMinHash (scale 10)
import ctypes as C  # noqa: E402
for name, n, p in (("f4 sets of MinimizerSketch k=21 w=11, 10M reads", 10_000_000, eng.params(L.MINIMIZER, 21, w=11)),
                   ("f4 sets of SyncmerSketch k=31 s=11, 12.5M reads", 12_500_000, eng.params(L.SYNCMER, 31, s=11))):
    b = eng.synth(L.ALPHA_DNA, n, 150, 0x5EED0003)
    res = eng.run(b, p)
    T = res.info()["n_tuples"]
    for scope, sname in ((L.SETS_PER_SEQUENCE, "per sequence"), (L.SETS_WHOLE_BATCH, "whole batch")):
        for scale in (1, 10):
            best, nv = 1e9, C.c_uint64()
            for _ in range(3):
                h = C.c_void_p()
                eng.lib.bsk_ctx_sync(eng.ctx)
                t = time.perf_counter()
                eng._chk(eng.lib.bsk_result_sets(eng.ctx, res.h, scope, scale, C.byref(h)))
                best = min(best, time.perf_counter() - t)
                eng.lib.bsk_sets_info(h, None, C.byref(nv))
                eng.lib.bsk_sets_release(h)
            row = {"case": f"{name}, {sname}, scale {scale}", "tuples_in": T, "values_out": nv.value, "call_ms": round(best * 1e3, 3),
                   "G_tuples_per_s": round(T / best / 1e9, 2), "algorithmic_GB": round((8 * T + 8 * nv.value + 16 * n) / 1e9, 3),
                   "achieved_GBps": round((8 * T + 8 * nv.value + 16 * n) / best / 1e9, 1)}
            rows.append(row)
            print(json.dumps(row), flush=True)
    del res, b
if len(sys.argv) > 1:
    json.dump({"rows": rows, "note": "units = bases (DNA) or residues (protein); device-resident input/output"},
              open(sys.argv[1], "w"), indent=1)
