"""GPU: the minimizer path off its headline point -- one JSON line per case.

  uniform read lengths 100 / 150 / 151 / 200 / 250 / 300 / 350 (and 150..350 through the syncmer kernels, k = 31 s = 11), a ragged batch (lengths uniform in 60..150: trimmed reads) with and
  without length-binned units, 2 % / 10 % of the reads ending in a 50-base poly-A tail.  k = 21, w = 11, ~3e9 bases per case,
  inputs and outputs resident in HBM, min / median of 5 launches (HIP events around the kernels, bsk_sketch_timed).
usage: python scripts/robustness_sweep.py [bases] > profiles/r04/robustness.jsonl"""
import json, os, sys
os.environ.setdefault("BSK_PY_WATCH_ENV", "1")  # this script flips BSK_* switches between runs
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bio_amd import sketches as S, _lib as L

BASES = float(sys.argv[1]) if len(sys.argv) > 1 else 3e9
ONLY_OUTLIERS = "--only-outliers" in sys.argv           # the class-plan rows alone
N_OUTLIER = int(float(os.environ.get("BSK_SWEEP_OUTLIER_READS", "0")))  # reads of the outlier cases (0: BASES / 150 / 1.5; the review asks for a 10^8-scale batch)
K, W = 21, 11
eng = S.Engine(0)
rng = np.random.default_rng(12)


def run(case, b, nbases, extra=None, p=None):
    p = p or eng.params(L.MINIMIZER, K, w=W)
    prep_first = eng.prepare(b, p)  # (allocates the view's buffers: a streaming caller re-uses them, bsk_batch_refill_ascii)
    res, ms = eng.run_timed(b, p, 2, 5)
    prep = min(eng.prepare(b, p) for _ in range(3)) if prep_first else 0.0  # (right behind the launches: an idle board clocks down, and the pass takes 3x as long)
    ms = sorted(ms)
    inf, plan = res.info(), res.plan()
    dg = res.digest()
    nparts, cut_ms = res.class_plan()  # (class plans: the pass that cuts the batch by length is paid by every bsk_sketch; eng.prepare above timed it
    # again right behind the launches -- cut_ms is the first, cold run's: an idle board clocks down and the pass takes several times as long)
    out = dict(case=case, kernel=plan["kernel"], waves_per_cu=plan["waves_per_cu"], reads=inf["n_reads"], bases=int(nbases), tuples=inf["n_tuples"],
               kernel_ms_min=round(ms[0], 4), kernel_ms_median=round(ms[len(ms) // 2], 4), gbases_per_s=round(nbases / ms[0] / 1e6, 1),
               gbases_per_s_median=round(nbases / ms[len(ms) // 2] / 1e6, 1), prepare_ms=round(prep, 4), prepare_first_ms=round(prep_first, 4),
               gbases_per_s_with_prepare=round(nbases / (ms[0] + prep) / 1e6, 1), checksum=dg["checksum"], first_window_tie_reads=dg.get("first_window_tie", None))
    if nparts:
        out.update(class_parts=nparts, class_cut_ms=round(cut_ms, 4))
    if extra:
        out.update(extra)
    print(json.dumps(out), flush=True)
    res.close()
    return out


def ragged(lo, hi, n):
    lens = rng.integers(lo, hi + 1, n, dtype=np.uint64)
    offs = np.zeros(n + 1, np.uint64)
    np.cumsum(lens, out=offs[1:])
    data = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(offs[-1]), dtype=np.uint8)]
    return data, offs


base = None
for rl in (() if ONLY_OUTLIERS else (100, 150, 151, 200, 250, 300, 350, 400, 500, 1000)):
    n = int(BASES / rl)
    b = eng.synth(L.ALPHA_DNA, n, rl, 0x5EED0003)
    o = run("uniform %d bp" % rl, b, n * rl)
    if rl == 150:
        base = o["gbases_per_s"]
    b.close()

ps = eng.params(L.SYNCMER, 31, s=11)
for rl in (() if ONLY_OUTLIERS else (150, 200, 250, 300, 350)):  # syncmers k = 31 s = 11 over the read length: k_syncmer_pk, then k_syncmer_pkl (round 4; k_syncmer_fast before)
    n = int(BASES / rl)
    b = eng.synth(L.ALPHA_DNA, n, rl, 0x5EED0003)
    run("syncmers k=31 s=11, uniform %d bp" % rl, b, n * rl, p=ps)
    b.close()

if ONLY_OUTLIERS:
    n = 2048
else:
    n = int(BASES / 105)  # (the same bases as the uniform cases: the rate of a batch grows with its size, and the ratio below is against them)
data, offs = ragged(60, 150, n)
b = eng.batch_from_arrays(data, offs)
r1 = run("ragged 60..150 bp, length-binned units", b, int(offs[-1]), dict(vs_uniform_150=None))
os.environ["BSK_NO_BIN"] = "1"
r2 = run("ragged 60..150 bp, units in batch order (BSK_NO_BIN)", b, int(offs[-1]))
os.environ.pop("BSK_NO_BIN", None)
os.environ["BSK_NO_BIN_EARLY"] = "1"  # (round 4's way: the view is built by the first plan that wants it, in classes of the plan's block -- bsk_batch_prepare times that pass)
b2 = eng.batch_from_arrays(data, offs)
r3 = run("ragged 60..150 bp, length-binned units, view built per plan (BSK_NO_BIN_EARLY)", b2, int(offs[-1]))
b2.close()
del os.environ["BSK_NO_BIN_EARLY"]
assert r1["checksum"] == r2["checksum"], "binned and unbinned digests differ"
# the same trimmed reads through the syncmer kernels (k = 31, s = 11)
s1 = run("syncmers k=31 s=11, ragged 60..150 bp, length-binned units", b, int(offs[-1]), p=ps)
os.environ["BSK_NO_BIN"] = "1"
s2 = run("syncmers k=31 s=11, ragged 60..150 bp, units in batch order (BSK_NO_BIN)", b, int(offs[-1]), p=ps)
del os.environ["BSK_NO_BIN"]
assert s1["checksum"] == s2["checksum"], "binned and unbinned syncmer digests differ"
b.close()
del data, offs

# outliers (round 5, class plans): 150-base reads + a few longer ones -- an outlier costs its own bases (sketch.go:46), not the batch's plan
n = N_OUTLIER or int(BASES / 150 / 1.5)
uni = None
for case, outl in (("150 bp (host arrays)", []), ("150 bp + 0.01 % of 400 bases", [(1e-4, 400)]), ("150 bp + 0.01 % of 5000 bases", [(1e-4, 5000)]),
                   ("150 bp + 1 % of 250 bases", [(1e-2, 250)]), ("150 bp + 0.01 % of 400 + 0.01 % of 5000 + 1 % of 250 bases", [(1e-4, 400), (1e-4, 5000), (1e-2, 250)])):
    lens = np.full(n, 150, np.uint64)
    for frac, ln in outl:
        lens[rng.integers(0, n, int(n * frac))] = ln
    offs = np.zeros(n + 1, np.uint64)
    np.cumsum(lens, out=offs[1:])
    data = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(offs[-1]), dtype=np.uint8)]
    b = eng.batch_from_arrays(data, offs)
    o = run(case + ", class plans", b, int(offs[-1]))
    if not outl:
        uni = o["gbases_per_s"]
    else:
        os.environ["BSK_NO_CLASS"] = "1"
        o1 = run(case + ", one plan (BSK_NO_CLASS)", b, int(offs[-1]))
        del os.environ["BSK_NO_CLASS"]
        assert o["checksum"] == o1["checksum"] and o["tuples"] == o1["tuples"], "class plans and the one-plan run differ"
        print(json.dumps(dict(case="summary: " + case, class_plans_over_uniform_150=round(o["gbases_per_s"] / uni, 3),
                              with_cut_passes_over_uniform_150=round(o["gbases_per_s_with_prepare"] / uni, 3), one_plan_over_uniform_150=round(o1["gbases_per_s"] / uni, 3))), flush=True)
    b.close()
    del data, offs, lens

n, rl = int(BASES / 150 / 1.5), 150
for frac in (() if ONLY_OUTLIERS else (0.02, 0.10)):
    d = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n * rl, dtype=np.uint8)].copy().reshape(n, rl)
    d[rng.random(n) < frac, 100:] = ord("A")
    b = eng.batch_from_arrays(d.reshape(-1), np.arange(n + 1, dtype=np.uint64) * rl)
    run("%d %% of the reads end in a 50-base poly-A tail" % round(frac * 100), b, n * rl)
    b.close()
if not ONLY_OUTLIERS:
  print(json.dumps(dict(case="summary", uniform_150=base, ragged_binned_over_uniform_150=round(r1["gbases_per_s"] / base, 3),
                      ragged_binned_with_prepare_over_uniform_150=round(r1["gbases_per_s_with_prepare"] / base, 3),
                      ragged_unbinned_over_uniform_150=round(r2["gbases_per_s"] / base, 3),
                      ragged_view_per_plan_over_uniform_150=round(r3["gbases_per_s"] / base, 3),
                      ragged_view_per_plan_with_prepare_over_uniform_150=round(r3["gbases_per_s_with_prepare"] / base, 3))))
