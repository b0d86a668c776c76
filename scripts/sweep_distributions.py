"""GPU: realistic read-length distributions (tests/distributions.py) at sweep size, one JSON line per (distribution, kind) -- appended to
profiles/r06/robustness_r06.jsonl.  rate = bases / best of 5 launches (HIP events); weighted_uniform = bases / sum_i(bases_i / rate_uniform(
nearest anchor length)), anchors measured in this session at the same total bases.  usage: python scripts/sweep_distributions.py [bases]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bio_amd import sketches as S, _lib as L
from tests import distributions as D

BASES = float(sys.argv[1]) if len(sys.argv) > 1 else 3e9
ANCHORS = (100, 151, 200, 250, 300, 1000, 8000, 30000)
eng = S.Engine(0)
kinds = (("minimizer k=21 w=11", eng.params(L.MINIMIZER, 21, w=11)), ("syncmer k=31 s=11", eng.params(L.SYNCMER, 31, s=11)))


def rate(b, p, nbases):
    res, ms = eng.run_timed(b, p, 2, 5)
    plan = res.plan()["kernel"]
    res.close()
    return nbases / min(ms) / 1e6, plan


uni = {}
for kname, p in kinds:
    for rl in ANCHORS:
        n = int(BASES / rl)
        b = eng.synth(L.ALPHA_DNA, n, rl, 0x5EED0D00 + rl)
        uni[(kname, rl)] = rate(b, p, n * rl)
        b.close()
        print(json.dumps(dict(case="distribution anchors: uniform %d bases, %s" % (rl, kname), gbases_per_s=round(uni[(kname, rl)][0], 1), kernel=uni[(kname, rl)][1])), flush=True)
for name in D.NAMES:
    rng = np.random.default_rng(abs(hash(name)) % (1 << 32))
    ln = D.lengths(name, BASES, rng)
    data, offs = D.batch_arrays(ln, rng)
    nbases = int(offs[-1])
    b = eng.batch_from_arrays(data, offs)
    anchors = np.array(ANCHORS)
    nearest = anchors[np.abs(np.log(ln[:, None].astype(np.float64)) - np.log(anchors[None, :])).argmin(axis=1)]
    for kname, p in kinds:
        got, plan = rate(b, p, nbases)
        t = sum(float(ln[nearest == a].sum()) / (uni[(kname, int(a))][0] * 1e9) for a in anchors)
        want = nbases / t / 1e9
        print(json.dumps(dict(case="distribution %s, %s" % (name, kname), reads=int(len(ln)), bases=nbases, gbases_per_s=round(got, 1),
                              weighted_uniform_gbases_per_s=round(want, 1), of_weighted_uniform=round(got / want, 3), kernel=plan)), flush=True)
    b.close()
