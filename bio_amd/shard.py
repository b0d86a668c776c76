"""Multi-GPU plumbing for the sketches path: shard by record, gather counters.

Reads are independent units (an iterator holds one sequence: iterator.go:61, sketch.go:46), so
the path shards with NO data-path collective: rank g owns the contiguous record range
[g*N/G, (g+1)*N/G) and its tuples stay on its GPU.  The only collective is one all_gather of a
few counters per rank (RCCL over xGMI with backend "nccl"; "gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple


def shard_range(n_records: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous record range [lo, hi) of `rank`; ranges tile [0, n) exactly, sizes differ by at most 1."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(n_records, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_counters(values: Sequence[float], device=None) -> List[List[float]]:
    """all_gather a small vector of per-rank counters; returns one list per rank (rank order).

    Works without an initialised process group (world 1) so single-GPU runs need no rendezvous.
    """
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [list(map(float, values))]
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[float(x) for x in o.tolist()] for o in out]


def whole_job(per_rank: List[List[float]], steps: int) -> Dict[str, float]:
    """per_rank rows are [seconds, bases, tuples]; whole-job throughput = total bases*steps / MAX seconds."""
    dt = max(r[0] for r in per_rank)
    bases = sum(r[1] for r in per_rank)
    tuples = sum(r[2] for r in per_rank)
    return {"seconds": dt, "bases": bases, "tuples": tuples, "gbases_per_s": bases * steps / dt / 1e9}
