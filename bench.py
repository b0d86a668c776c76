#!/usr/bin/env python3
"""bench.py -- headline benchmark of the sketches/ hot path on MI355X.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on):
    100M x 150 bp synthetic reads, minimizer sketch k=21 w=11, canonical ntHash.
One "step" = one pass of the minimizer kernel over the whole device-resident batch
(2-bit packed reads in HBM -> (hash, pos|strand) tuples + per-read index in HBM).
Reads shard by record: with N GPUs every rank owns its own 100M-read batch (weak
scaling, no data-path collective); the only collective is one RCCL all_gather of the
per-rank counters at the end.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for the fields).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling

WORKLOADS = {
    # name: (kind, n_reads, read_len, k, w_or_s, description)
    "minimizer": ("min", 100_000_000, 150, 21, 11, "100M x 150 bp reads, minimizer sketch k=21 w=11 (BASELINE configs[2])"),
    "nthash": ("nt", 10_000_000, 150, 21, 0, "10M x 150 bp reads, canonical ntHash stream k=21 (BASELINE configs[1])"),
    "syncmer": ("syn", 125_000_000, 150, 31, 11, "125M x 150 bp reads per GPU (1B over 8), syncmer sketch k=31 s=11 (BASELINE configs[3])"),
    "protmin": ("pmin", 50_000_000, 300, 9, 5, "50M x 300 aa, protein minimizer sketch k=9 w=5 (BASELINE configs[4])"),
    "kmer": ("kmer", 10_000_000, 150, 21, 0, "10M x 150 bp reads, canonical 2-bit k-mer codes k=21"),
    "prothash": ("phash", 20_000_000, 300, 9, 0, "20M x 300 aa, protein k-mer hashes k=9"),
    "simhash": ("sim", 20_000_000, 150, 21, 5, "20M x 150 bp reads, SimHash k=21 m=5 scale=5"),
}
KERNELS = {"min": "k_minimizer_fast<11,32,true>", "nt": "k_nthash_fast<1>", "syn": "k_syncmer_fast<20>", "pmin": "k_prot_minimizer_fast<5,9>",
           "kmer": "k_nthash_fast<2>", "phash": "k_prot_hash_fast<9>", "sim": "k_simhash_fast<5,12>"}
NOTES = {
    "min": "integer-VALU bound, not HBM bound (DESIGN.md 3.1); frac is vs the 8 TB/s spec peak",
    "nt": "HBM-write bound (DESIGN.md 3.2); a plain 16 B/lane fill kernel reaches 5.2-5.9 TB/s on this part",
    "syn": "integer-VALU bound (two rolling hashes + a 2(k-s) window per base; DESIGN.md 3.3)",
    "pmin": "integer-VALU bound (wyhash from scratch per residue: 8 v_mad_u64_u32; DESIGN.md 3.4)",
    "kmer": "HBM-write bound, same streaming kernel as ntHash (DESIGN.md 3.2)",
    "phash": "HBM-write bound (DESIGN.md 3.4)",
    "sim": "integer-VALU bound (two rolling hashes + bit-sliced counters, ~110 ops per k-mer; DESIGN.md 3.6)",
}
ORACLE_KIND = {"min": 4, "nt": 2, "syn": 5, "pmin": 7, "kmer": 1, "phash": 6, "sim": 3}
PROTEIN = ("pmin", "phash")
STREAM = ("nt", "kmer", "phash", "sim")


def cpu_baseline(kind: str, k: int, x: int, read_len: int, seed: int):
    """Oracle (a CPU restatement of the reference algorithm, NOT the Go binary) on the host cores.

    Bounded sample of the same synthetic workload; the reference's per-read iterator
    state machine (sorted buffer, binary-search insert), one iterator per read, OpenMP over reads.
    """
    import numpy as np
    from oracle import oracle as O
    cores = len(os.sched_getaffinity(0))
    rng = np.random.default_rng(seed)
    okind = ORACLE_KIND[kind]
    letters = b"ACDEFGHIKLMNPQRSTVWY" if kind in PROTEIN else b"ACGT"

    def run(n, threads):
        data = np.frombuffer(letters, np.uint8)[rng.integers(0, len(letters), n * read_len)]
        offs = (np.arange(n + 1, dtype=np.uint64) * read_len)
        O.batch_run(okind, data[: 1000 * read_len], offs[:1001], k, x, threads=threads)  # warm
        t = time.perf_counter()
        O.batch_run(okind, data, offs, k, x, threads=threads)
        return n * read_len / (time.perf_counter() - t) / 1e9

    scale = 150.0 / read_len
    v1 = run(int(200_000 * scale), 1)
    n_all = int(min(4_000_000, 400_000 * cores) * scale)
    vall = run(n_all, cores)
    return {
        "value": round(vall, 4), "unit": "Gbases/s", "cores": cores, "kind": "port",
        "value_1thread": round(v1, 4),
        "sample": f"{n_all} synthetic {read_len}-letter sequences on {cores} threads (and {int(200_000 * scale)} on 1 thread); "
                  "C restatement of the reference state machine (oracle/bio_oracle.c), not the Go binary",
    }


def measured_traffic(workload: str, n_reads: int):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*/traffic.json), or None.

    bench.py cannot run the profiler on itself; the counters were collected on this same command in separate
    --pmc passes (FETCH_SIZE, WRITE_SIZE) and corrected as MI355X_MICROARCH.md prescribes (KiB units, FETCH x2).
    """
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for d in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        f = os.path.join(pdir, d, "traffic.json")
        if os.path.exists(f):
            for e in json.load(open(f)).get("entries", []):
                if e.get("workload") == workload and e.get("reads_per_gpu") == n_reads:
                    best = e.get("hbm_bytes_per_launch")
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="minimizer", choices=sorted(WORKLOADS))
    ap.add_argument("--reads", type=float, default=0, help="override reads per GPU (dev)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the sketch engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from bio_amd import _lib as L
    from bio_amd import sketches as S

    kind, n_reads, read_len, k, x, desc = WORKLOADS[args.workload]
    if args.reads:
        n_reads = int(args.reads)
    seed = 0x5EED0000 + 3 + 0x1000000 * rank  # each rank hashes its own shard of the synthetic stream
    eng = S.Engine(local_rank)
    batch = eng.synth(L.ALPHA_PROTEIN if kind in PROTEIN else L.ALPHA_DNA, n_reads, read_len, seed)
    p = {"min": lambda: eng.params(L.MINIMIZER, k, w=x), "nt": lambda: eng.params(L.NTHASH, k), "syn": lambda: eng.params(L.SYNCMER, k, s=x),
         "pmin": lambda: eng.params(L.PROT_MINIMIZER, k, w=x), "kmer": lambda: eng.params(L.KMER, k),
         "phash": lambda: eng.params(L.PROT_HASH, k), "sim": lambda: eng.params(L.SIMHASH, k, m=x, scale=5)}[kind]()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed: first run sizes the result buffers; then W warm-up steps
    res = eng.run(batch, p)
    res, _ = eng.run_timed(batch, p, args.warmup, 0, reuse=res)
    barrier()
    t0 = time.perf_counter()
    res, kernel_ms = eng.run_timed(batch, p, 0, args.steps, reuse=res)  # exactly K steps, HIP events around each kernel
    barrier()
    dt = time.perf_counter() - t0

    info = res.info()
    tuples = info["n_tuples"]
    # whole-job numbers: MAX time over ranks, SUM of units over ranks (one RCCL all_gather of counters)
    from bio_amd.shard import gather_counters, whole_job
    job = whole_job(gather_counters([dt, float(n_reads * read_len), float(tuples)], device=dev), args.steps)
    dt_max, bases_total, tuples_total = job["seconds"], job["bases"], job["tuples"]

    if rank == 0:
        ms_per_step = dt_max / args.steps * 1e3
        value = bases_total * args.steps / dt_max / 1e9
        # algorithmic bytes per launch on THIS rank: packed bases + one u64 descriptor in,
        # 12 B per tuple (u64 hash + u32 pos|strand) + one u64 index word per read out.
        in_bytes = n_reads * ((read_len + 8) if kind in PROTEIN else ((read_len + 3) // 4 + 8))  # residues are bytes, bases 2 bits
        if kind in STREAM:
            alg_bytes = in_bytes + 8 * tuples  # positions and offsets implicit
        else:
            alg_bytes = in_bytes + 12 * tuples + 8 * n_reads
        k_ms = sum(kernel_ms) / len(kernel_ms)
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        unit = "Gresidues/s" if kind in PROTEIN else "Gbases/s"
        metric = {"min": "Gbases/s hashed (k=21 ntHash + minimizer)", "nt": "Gbases/s hashed (k=21 ntHash stream)",
                  "syn": "Gbases/s hashed (k=31 s=11 syncmer)", "pmin": "Gresidues/s hashed (k=9 w=5 protein minimizer)",
                  "kmer": "Gbases/s encoded (k=21 canonical k-mer codes)", "phash": "Gresidues/s hashed (k=9 wyhash)",
                  "sim": "Gbases/s hashed (k=21 m=5 SimHash)"}[kind]
        par = {"min": ("w", x), "syn": ("s", x), "pmin": ("w", x), "sim": ("m", x)}.get(kind, ("canonical", True))
        out = {
            "metric": metric,
            "value": round(value, 2), "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": desc, "reads_per_gpu": n_reads, "read_len": read_len, "k": k, par[0]: par[1],
                       "tuples_per_gpu": int(tuples), "tuples_total": int(tuples_total),
                       "parallelism": f"reads sharded by record over {world} GPU(s), no data-path collective",
                       "input": ("residues (1 B each)" if kind in PROTEIN else "2-bit packed reads") + " resident in HBM",
                       "output": ("hash u64 per position" if kind in STREAM else "hash u64 + pos|strand u32") + " + u64 index per read, in HBM"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": measured_traffic(args.workload, n_reads),
                         "kernel": KERNELS[kind],
                         "kernel_ms_avg": round(k_ms, 4), "algorithmic_bytes_per_launch": int(alg_bytes),
                         "read_only_frac": round(in_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "note": NOTES[kind] + "; read_only_frac = input bytes alone over the same peak (north_star's 'HBM-read roofline')"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(kind, k, x, read_len, 12345)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
