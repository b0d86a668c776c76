"""GPU: the planner off the data it was fitted on.  Every threshold of make_plan_enc / tile_positions / class plans was
measured on uniform random reads of ONE length; a real run folder is not that.  Three length distributions (tests/distributions.py) must
keep at least 0.6 of their LENGTH-WEIGHTED UNIFORM rate -- bases / sum_i(bases_i / rate_uniform(len_i)), the uniform rates measured in the
same session at anchor lengths -- for minimizers (k=21 w=11) and syncmers (k=31 s=11), whatever mix of kernels, class plans, length-binned
units and tiles the planner picks; and sampled reads must equal the oracle's (the plan may be anything, the tuples may not)."""
import numpy as np
import pytest

from bio_amd import _lib as L
from tests import distributions as D

pytestmark = pytest.mark.gpu

ANCHORS = (100, 151, 200, 250, 300, 1000, 8000, 30000)
BASES = 4e8


def rate(engine, b, p, nbases):
    res, ms = engine.run_timed(b, p, 1, 3)
    plan = res.plan()["kernel"]
    res.close()
    return nbases / min(ms) / 1e6, plan


@pytest.fixture(scope="module")
def uniform_rates(engine):
    out = {}
    for kind, pk in (("min", dict(k=21, w=11)), ("syn", dict(k=31, s=11))):
        p = engine.params(L.MINIMIZER if kind == "min" else L.SYNCMER, **pk)
        for rl in ANCHORS:
            n = max(int(BASES / rl), 2000)
            b = engine.synth(L.ALPHA_DNA, n, rl, 0x5EED0D00 + rl)
            out[(kind, rl)] = rate(engine, b, p, n * rl)[0]
            b.close()
    return out


@pytest.mark.parametrize("name", D.NAMES)
def test_distribution_keeps_its_length_weighted_uniform_rate(engine, oracle, uniform_rates, name):
    rng = np.random.default_rng(abs(hash(name)) % (1 << 32))
    ln = D.lengths(name, BASES, rng)
    data, offs = D.batch_arrays(ln, rng)
    nbases = int(offs[-1])
    b = engine.batch_from_arrays(data, offs)
    anchors = np.array(ANCHORS)
    nearest = anchors[np.abs(np.log(ln[:, None].astype(np.float64)) - np.log(anchors[None, :])).argmin(axis=1)]
    report = {}
    for kind, pk, fn in (("min", dict(k=21, w=11), lambda q: oracle.minimizer(q, 21, 11, False, closed=True)),
                         ("syn", dict(k=31, s=11), lambda q: oracle.syncmer(q, 31, 11, False, closed=True))):
        p = engine.params(L.MINIMIZER if kind == "min" else L.SYNCMER, **pk)
        got, plan = rate(engine, b, p, nbases)
        t = sum(float(ln[nearest == a].sum()) / (uniform_rates[(kind, int(a))] * 1e9) for a in anchors)
        want = nbases / t / 1e9
        report[kind] = (round(got, 1), round(want, 1), plan)
        assert got >= 0.6 * want, (name, kind, got, want, plan)
        res = engine.run(b, p)
        for i in list(range(0, len(ln), max(1, len(ln) // 60))) + [int(np.argmax(ln)), int(np.argmin(ln))]:
            q = data[int(offs[i]):int(offs[i + 1])].tobytes().decode()
            st, h, pos = res.read(i)
            try:
                eh, ep, es, fl = fn(q)
            except oracle.OracleError as e:
                assert e.name == "ErrShortSeq" and (st & L.ST_CODE_MASK) == L.ST_SHORT and len(h) == 0, (name, kind, i, len(q))
                continue
            assert np.array_equal(h, eh) and np.array_equal(pos & L.POS_MASK, ep) and np.array_equal(pos >> 31, es), (name, kind, i, len(q), plan)
        res.close()
    print(name, report)
    b.close()
