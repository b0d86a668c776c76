#!/usr/bin/env python3
"""Generates tests/golden/sketches_golden.json.

The reference (Go, un-vendored module deps, no Go toolchain here) cannot be run, so the
vectors are: (1) the reference's own known-answer values, copied as data from
sketches/sketch_test.go:67-72 (`ref_kat`), and (2) outputs of the CPU oracle
(oracle/bio_oracle.c), which reproduces (1), on the reference's inline test strings
(iterator_test.go:32,109-110; sketch_test.go:34,79) and on adversarial strings.
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

REF_STRINGS = {
    "sketch_test_minimizer": "GGCAAGTTCGTCA",                      # sketch_test.go:34
    "sketch_test_syncmer": "GGCAAGTTCGTCATCGATC",                  # sketch_test.go:79
    "iterator_test_100bp": "AAGTTTGAATCATTCAACTATCTAGTTTTCAGAGAACAATGTTCTCTAAAGAATAGAAAAGAGTCATTGTGCGGTGATGATGGCGGGAAGGATCCACCTG",
    "simhash_1": "GAACAATGTTCTCTAAAATTG",                          # iterator_test.go:109
    "simhash_2": "GcACAATGTTCTCTAAAATTG",                          # iterator_test.go:110
}
rng = random.Random(20250905)
ADV = {
    "homopolymer": "A" * 150,
    "period2": "AC" * 75,
    "period3": "ACG" * 50,
    "rc_palindrome": "ACGTTGCAACGT" * 12,
    "ecoRI_repeat": "GAATTC" * 25,
    "with_N": "".join(rng.choice("ACGTN") for _ in range(150)),
    "lowercase_mix": "".join(rng.choice("ACGTacgt") for _ in range(150)),
    "random150": "".join(rng.choice("ACGT") for _ in range(150)),
    "random31": "".join(rng.choice("ACGT") for _ in range(31)),
    "random30": "".join(rng.choice("ACGT") for _ in range(30)),
}


def ints(a):
    return [int(x) for x in a]


def entry(fn, *args, **kw):
    try:
        r = fn(*args, **kw)
    except O.OracleError as e:
        return {"error": e.name}
    if isinstance(r, tuple):
        return [ints(x) if hasattr(x, "__len__") else int(x) for x in r]
    return ints(r)


def main():
    g = {"ref_kat": {"seq": "GGCAAGTTCGTCA", "k": 5, "w": 3,
                     "codes": [973456138564179607, 2645801399420473919, 1099502864234245338, 6763474888237448943,
                               2737971715116251183],
                     "source": "sketches/sketch_test.go:67-72 (TestMinimizer)"},
         "cases": []}
    seqs = dict(REF_STRINGS)
    seqs.update(ADV)
    for name, s in seqs.items():
        for k, w in [(5, 3), (21, 11), (31, 15), (10, 1)]:
            g["cases"].append({"name": name, "seq": s, "fn": "minimizer", "k": k, "w": w,
                               "out": entry(O.minimizer, s, k, w)})
        for k, sm in [(5, 2), (31, 11), (31, 16), (7, 7)]:
            g["cases"].append({"name": name, "seq": s, "fn": "syncmer", "k": k, "s": sm,
                               "out": entry(O.syncmer, s, k, sm)})
        for k, canon, circ in [(10, True, False), (21, True, False), (21, False, False), (5, True, True)]:
            g["cases"].append({"name": name, "seq": s, "fn": "nthash", "k": k, "canonical": canon, "circular": circ,
                               "out": entry(O.nthash, s, k, canon, circ)})
        for k, canon in [(10, True), (10, False), (31, True)]:
            g["cases"].append({"name": name, "seq": s, "fn": "kmer", "k": k, "canonical": canon,
                               "out": entry(O.kmer_codes, s, k, canon)})
        g["cases"].append({"name": name, "seq": s, "fn": "simhash", "k": 21, "m": 5, "scale": 5,
                           "out": entry(O.simhash, s, 21, 5, 5)})
    prot = {"prot_all20x2": "ACDEFGHIKLMNPQRSTVWY" * 2,
            "prot_random": "".join(rng.choice("ACDEFGHIKLMNPQRSTVWY") for _ in range(120)),
            "prot_lowcomplex": "AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA"}
    for name, s in prot.items():
        for k, w in [(9, 5), (10, 3), (3, 1)]:
            g["cases"].append({"name": name, "seq": s, "fn": "protein_minimizer", "k": k, "w": w,
                               "out": entry(O.protein_minimizer, s, k, w)})
        g["cases"].append({"name": name, "seq": s, "fn": "protein_hashes", "k": 9, "out": entry(O.protein_hashes, s, 9)})
    # wyhash (github.com/zeebo/wyhash v0.0.1, seed 1 as at iterator-protein.go:87) on its own: every length class of the algorithm
    for n in (0, 1, 3, 4, 7, 8, 9, 15, 16, 17, 24, 31, 32, 33, 48, 63, 64, 65, 100):
        data = "".join(rng.choice("ACDEFGHIKLMNPQRSTVWY") for _ in range(n))
        for seed in (1, 0):
            g["cases"].append({"name": "wyhash_len%d" % n, "seq": data, "fn": "wyhash", "seed": seed, "out": [O.wyhash(data, seed)]})
    # ties inside the first sorted window (sorts.Quicksort is unstable, sketch.go:236,351): random short-alphabet strings whose oracle
    # result carries BSK_ST_FIRST_WINDOW_TIE (0x10) -- the pin harness says what upstream really yields for them
    found = 0
    for trial in range(20000):
        s_ = "".join(rng.choice("AC") for _ in range(rng.randint(24, 60)))
        k, w = rng.choice([(4, 6), (5, 4), (3, 8), (6, 5)])
        r = O.minimizer(s_, k, w)
        if r[3] & 0x10 and len(set(s_)) > 1:
            g["cases"].append({"name": "first_window_tie_%d" % found, "seq": s_, "fn": "minimizer", "k": k, "w": w, "out": entry(O.minimizer, s_, k, w)})
            ks, ss = rng.choice([(6, 3), (7, 4), (5, 2)])
            g["cases"].append({"name": "first_window_tie_%d" % found, "seq": s_, "fn": "syncmer", "k": ks, "s": ss, "out": entry(O.syncmer, s_, ks, ss)})
            found += 1
            if found == 8:
                break
    # k > 64 (Go's << by 64 or more yields 0: does nthash v0.4.0 rotate modulo 64?)
    long_ = "".join(rng.choice("ACGT") for _ in range(200))
    for k in (64, 65, 70, 100):
        g["cases"].append({"name": "k_over_64", "seq": long_, "fn": "nthash", "k": k, "canonical": True, "circular": False,
                           "out": entry(O.nthash, long_, k, True, False)})
        g["cases"].append({"name": "k_over_64", "seq": long_, "fn": "minimizer", "k": k, "w": 5, "out": entry(O.minimizer, long_, k, 5)})
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sketches_golden.json")
    with open(out, "w") as f:
        json.dump(g, f, separators=(",", ":"))
    print("wrote", out, len(g["cases"]), "cases", os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
