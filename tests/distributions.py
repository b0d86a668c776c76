"""Realistic read-length distributions for the planner checks (tests/test_gpu_distributions.py, scripts/robustness_sweep.py): none of them
is what the planner's thresholds were fitted on (uniform random reads of one length)."""
import numpy as np


def lengths(name: str, n_bases: float, rng) -> np.ndarray:
    if name == "illumina_2x151_trimmed":  # 2 x 151 with 5 % of the reads adapter-trimmed to 35..150 bases
        n = int(n_bases / 147)
        ln = np.full(n, 151, np.uint64)
        t = rng.random(n) < 0.05
        ln[t] = rng.integers(35, 151, int(t.sum()))
    elif name == "miseq_250_300_mix":  # a MiSeq run folder: 2 x 250 and 2 x 300 kits, a tenth of each quality-trimmed by up to a third
        n = int(n_bases / 265)
        ln = np.where(rng.random(n) < 0.5, 250, 300).astype(np.uint64)
        t = rng.random(n) < 0.10
        ln[t] = (ln[t] * (1.0 - rng.random(int(t.sum())) / 3.0)).astype(np.uint64)
    elif name == "lognormal_long_8kb":  # long reads: log-normal, median 8 kb, sigma 0.6, between 500 bases and 100 kb
        n = int(n_bases / 9600)
        ln = np.clip(rng.lognormal(np.log(8000.0), 0.6, n), 500, 100000).astype(np.uint64)
    else:
        raise ValueError(name)
    return ln


NAMES = ("illumina_2x151_trimmed", "miseq_250_300_mix", "lognormal_long_8kb")


def batch_arrays(ln: np.ndarray, rng):
    offs = np.zeros(len(ln) + 1, np.uint64)
    np.cumsum(ln, out=offs[1:])
    data = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(offs[-1]), dtype=np.uint8)]
    return data, offs
