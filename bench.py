#!/usr/bin/env python3
"""bench.py -- headline benchmark of the sketches/ hot path on MI355X.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on):
    100M x 150 bp synthetic reads, minimizer sketch k=21 w=11, canonical ntHash.
One "step" = one pass of the minimizer kernel over the whole device-resident batch
(2-bit packed reads in HBM -> (hash, pos|strand) tuples + per-read index in HBM).
Reads shard by record: with N GPUs every rank owns its own 100M-read batch (weak
scaling, no data-path collective); the only collective is one RCCL all_gather of the
per-rank counters at the end, issued through the library's own C ABI (bsk_gather_counts).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus 8                      # no launcher in the environment: bench.py starts its own 8 ranks
    python bench.py --gpus 8 --workload syncmer   # BASELINE configs[3]: 1B x 150 bp over 8 GPUs (125M reads per rank), k=31 s=11
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

With --gpus N > 1 and no launcher environment (RANK / WORLD_SIZE unset) the script re-executes itself through
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port>`: one rank per GPU,
the same line on rank 0.  A line never says n_gpus: 1 for --gpus N: a WORLD_SIZE that disagrees with --gpus is an error.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for the fields).

`--backend gloo --plumbing-only` (tests/test_shard_gloo.py) runs the N > 1 control flow -- rendezvous, barriers, counter
gather, whole-job arithmetic, the JSON line -- on CPUs with fixed stand-in counters and NO kernel; its line says so and is not
a measurement.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_COPY_GBS = 6290.0   # measured float4-copy ceiling on this part (same guide)

WORKLOADS = {
    # name: (kind, n_reads, read_len, k, w_or_s, description)
    "minimizer": ("min", 100_000_000, 150, 21, 11, "100M x 150 bp reads, minimizer sketch k=21 w=11 (BASELINE configs[2])"),
    "nthash": ("nt", 10_000_000, 150, 21, 0, "10M x 150 bp reads, canonical ntHash stream k=21 (BASELINE configs[1])"),
    "syncmer": ("syn", 125_000_000, 150, 31, 11, "125M x 150 bp reads per GPU (1B over 8), syncmer sketch k=31 s=11 (BASELINE configs[3])"),
    "protmin": ("pmin", 50_000_000, 300, 9, 5, "50M x 300 aa, protein minimizer sketch k=9 w=5 (BASELINE configs[4])"),
    "kmer": ("kmer", 10_000_000, 150, 21, 0, "10M x 150 bp reads, canonical 2-bit k-mer codes k=21"),
    "prothash": ("phash", 20_000_000, 300, 9, 0, "20M x 300 aa, protein k-mer hashes k=9"),
    "simhash": ("sim", 20_000_000, 150, 21, 5, "20M x 150 bp reads, SimHash k=21 m=5 scale=5"),
    # off the headline point (the same 1.5e10 bases): the unit-row kernel's length range (scripts/robustness_sweep.py has the rest)
    "minimizer250": ("min", 60_000_000, 250, 21, 11, "60M x 250 bp reads, minimizer sketch k=21 w=11 (the configs[2] parameters on longer reads)"),
    "minimizer400": ("min", 37_500_000, 400, 21, 11, "37.5M x 400 bp reads, minimizer sketch k=21 w=11 (the configs[2] parameters beyond the unit-row kernel's reach: k_minimizer_pkd, DESIGN.md 3.1c)"),
    "syncmer250": ("syn", 60_000_000, 250, 31, 11, "60M x 250 bp reads, syncmer sketch k=31 s=11 (the configs[3] parameters on longer reads: k_syncmer_pfl, DESIGN.md 3.3)"),
}
NOTES = {
    "min": "bound by the in-order instruction issue of two waves per SIMD (eight waves per CU: LDS staging) with the board's power cap as a second ceiling 5-8 % "
           "above it -- the three-wave kernel (k_minimizer_ring) needs 8 % fewer cycles and is clocked 9 % lower; `power` holds this run's board watts and clock "
           "(every kernel of the library draws 1.26-1.38 kW of the 1.4 kW cap; DESIGN.md 3.1) -- not by HBM; the VALU pipe is ~0.6 full",
    "nt": "HBM-write bound (DESIGN.md 3.2); a plain 16 B/lane fill kernel reaches 5.2-5.9 TB/s on this part",
    "syn": "integer-VALU bound: k_syncmer_pf -- the rolling s-mer hash + a 2(k-s) window per base, then the ~7 selected k-mers of a read hashed from scratch at the end of every unit (round 6; DESIGN.md 3.3)",
    "pmin": "integer-VALU bound (wyhash from scratch per residue: 8 v_mad_u64_u32; DESIGN.md 3.4)",
    "kmer": "HBM-write bound, same streaming kernel as ntHash (DESIGN.md 3.2)",
    "phash": "HBM-write bound (DESIGN.md 3.4)",
    "sim": "integer-VALU bound (two rolling hashes + bit-sliced counters, ~110 ops per k-mer; DESIGN.md 3.6)",
}
ORACLE_KIND = {"min": 4, "nt": 2, "syn": 5, "pmin": 7, "kmer": 1, "phash": 6, "sim": 3}
PROTEIN = ("pmin", "phash")
STREAM = ("nt", "kmer", "phash", "sim")


def cpu_baseline(kind: str, k: int, x: int, read_len: int, batch):
    """Oracle (a CPU restatement of the reference algorithm, NOT the Go binary) on the host cores.

    A bounded sample of THE SAME synthetic batch the GPU hashes (its first reads, decoded back to ASCII): the reference's
    per-read iterator state machine (sorted buffer, binary-search insert), one recycled iterator per thread as sync.Pool gives
    a Go worker, reads borrowed not copied, OpenMP over reads.  Thread-scaling row: 1 / 16 / 64 / all threads.
    """
    from oracle import oracle as O
    cores = len(os.sched_getaffinity(0))
    okind = ORACLE_KIND[kind]
    scale = 150.0 / read_len
    threads = sorted({t for t in (1, 16, 64, cores) if t <= cores})
    n_max = int(min(4_000_000, 200_000 * cores) * scale)
    n_max = min(n_max, batch.info()["n_reads"])
    data, offs = batch.fetch_ascii(0, n_max)
    O.batch_run(okind, data[: int(offs[1000])], offs[:1001], k, x, threads=1)  # warm
    rows = []
    for t in threads:
        n = min(n_max, int(200_000 * t * scale))
        t0 = time.perf_counter()
        O.batch_run(okind, data, offs[: n + 1], k, x, threads=t)
        dt = time.perf_counter() - t0
        rows.append({"threads": t, "reads": n, "value": round(n * read_len / dt / 1e9, 4)})
    quota = None
    try:
        quota = open("/sys/fs/cgroup/cpu.max").read().split()
    except OSError:
        pass
    quota_cores = None
    if quota and quota[0] != "max":
        quota_cores = float(quota[0]) / float(quota[1])
    best = max(rows, key=lambda r: r["value"])
    v1, vall = rows[0]["value"], best["value"]
    smt = 1
    try:
        sib = open("/sys/devices/system/cpu/cpu0/topology/thread_siblings_list").read().strip()
        smt = max(1, len([p for p in sib.replace("-", ",").split(",") if p]))
    except OSError:
        pass
    phys = max(1, cores // smt)
    out = {
        "value": vall, "unit": "Gresidues/s" if kind in PROTEIN else "Gbases/s", "cores": best["threads"], "kind": "port",
        "value_1thread": v1, "thread_scaling": rows, "hardware_threads": cores, "physical_cores": phys,
        "speedup_best_over_1": round(vall / v1, 2) if v1 else None,
        "sample": f"the first {best['reads']} sequences of the GPU's own synthetic batch ({read_len} letters each) on {best['threads']} threads "
                  f"(the fastest row of thread_scaling); C restatement of the reference state machine (oracle/bio_oracle.c), pooled iterators, not the Go binary",
    }
    usable = min(float(phys), quota_cores) if quota_cores else float(phys)
    out["usable_cores"] = usable
    out["scaling_efficiency_vs_usable_cores"] = round(vall / v1 / usable, 3) if v1 else None
    if quota:
        out["cgroup_cpu_max"] = " ".join(quota)
        if quota_cores:
            out["note"] = (f"the box's cgroup grants {quota_cores:g} CPUs of time ({quota[0]}/{quota[1]} us): rows with more threads than that "
                           "are throttled, not faster; `cores` is the thread count of the fastest row")
    if "note" not in out and v1 and vall / v1 < 0.5 * phys:
        out["note"] = (f"{smt}-way SMT: {cores} hardware threads share {phys} cores, and ~300 B/read of short-lived iterator state "
                       "per thread keeps the run memory/allocator-bound well before all threads are busy")
    return out


def end_to_end(kind: str, p, batch, read_len: int, n_streams: int = 5):
    """Host bytes -> tuples on the host (bsk_pipeline_*; SURVEY 8d "reported separately, not the metric"): the same reads,
    decoded back to ASCII, go (a) from host memory and (b) from a plain / a gzip FASTQ (FASTA for protein) file through
    pinned chunks, H2D + pack, the kernel and the D2H of every tuple, with n_streams streams overlapping those stages.
    Bounded samples (a few seconds in all).  NEVER `value`."""
    import gzip
    import tempfile

    import numpy as np
    from bio_amd import sketches as S
    n_have = batch.info()["n_reads"]
    scale = 150.0 / read_len
    n_mem = min(n_have, int(16_000_000 * scale))
    data, offs = batch.fetch_ascii(0, n_mem)
    alpha = 1 if kind in PROTEIN else 0
    n_mem_streams = int(os.environ.get("BSK_BENCH_MEM_STREAMS", str(max(n_streams, 8))))  # (the memory path has no parser threads to share the 16 CPUs with)
    out = {"n_streams": n_streams, "n_streams_from_memory": n_mem_streams, "chunk_records": 1 << 18,
           "what": "host ASCII -> 2-bit packed into pinned chunks on the worker threads (protein / reads with another byte: ASCII + pack kernel) -> H2D -> kernel -> "
                   "every tuple back in pinned host memory (u32 offsets, u16 positions); stages of different chunks overlap"}
    S.Engine.pipeline_trim()
    st0 = S.Engine.pipeline_memory(data, offs, p, n_streams=n_mem_streams, chunk_records=1 << 18, repeat=2, fetch=True, alphabet=alpha)
    # the second call of the process: the pinned buffers come from the library's pool (a long-lived host pins once, not per file)
    st = S.Engine.pipeline_memory(data, offs, p, n_streams=n_mem_streams, chunk_records=1 << 18, repeat=2, fetch=True, alphabet=alpha)
    out["pinned_buffers"] = ("pooled by the library between pipeline calls; from_memory.first_call is the process's first run (it pins "
                             "%.2f s summed over its threads), every figure below ran with the pool warm" % st0["pin_seconds"])
    out["from_memory"] = {"first_call": {"value": round(st0["bases"] / st0["seconds"] / 1e9, 3), "seconds": round(st0["seconds"], 4), "pin_seconds": round(st0["pin_seconds"], 4)},
                          "value": round(st["bases"] / st["seconds"] / 1e9, 3), "unit": "Gresidues/s" if alpha else "Gbases/s",
                          "reads": st["records"], "seconds": round(st["seconds"], 4),
                          "stage_seconds_summed_over_streams": {k: round(st[k], 4) for k in ("reader_seconds", "reader_wait_seconds", "h2d_pack_seconds", "kernel_seconds", "fetch_seconds")},
                          "bound": "the workers' host passes (2-bit pack into pinned memory, one pass over the fetched tuples) and the D2H copies: "
                                   "%.0f B/read up (packed words + descriptor; ASCII chunks %.0f) + %.0f B/read of tuples down "
                                   "(DESIGN.md 4; r03 moved 158 + 282 and reached 16.7 Gbases/s)" % (
                                       (read_len + 3) // 4 + 8 if not alpha else read_len + 8, read_len + 8,
                                       ((10.0 if kind not in STREAM else 8.0) * st["tuples"] / max(st["records"], 1) + 5))}
    # the pipeline WITH a consumer (bsk_pipeline_open_memory / _next / _release: every chunk delivered in record order, the role of fastx's
    # ChunkChan, seqio/fastx/reader.go:562-608).  The consumer here touches every chunk's counts and one value per chunk -- a real one
    # does its own work on its own thread while the workers run ahead (2 x workers + 2 output buffers).
    def sink_run(sink, scale_, streams):
        seen = vals = link = recs = 0
        order_ok = True
        with S.Engine.pipeline_open(p, data=data, offsets=offs, devices=[0], n_streams=streams, chunk_records=1 << 18, sink=sink, sets_scale=scale_,
                                    alphabet=alpha, repeat=2) as pl:
            for c in pl.chunks():
                order_ok &= c.sequence == seen
                seen += 1
                recs += c.n_records
                vals += c.n_values
                link += c.link_bytes
                if c.n_values:
                    _ = int(c.hash[0]) + int(c.offsets[-1])
        stx = pl.stats
        return {"value": round(stx["bases"] / stx["seconds"] / 1e9, 3), "unit": "Gresidues/s" if alpha else "Gbases/s", "n_streams": streams, "reads": recs, "chunks": seen,
                "delivered_in_order": bool(order_ok), "values_delivered": vals, "d2h_bytes_per_read": round(link / max(recs, 1), 2),
                "h2d_bytes_per_read": (read_len + 3) // 4 + 8 if not alpha else read_len + 8, "seconds": round(stx["seconds"], 4),
                "stage_seconds_summed_over_streams": {k_: round(stx[k_], 4) for k_ in ("h2d_pack_seconds", "kernel_seconds", "fetch_seconds")}}
    try:
        from bio_amd import _lib as _L
        out["from_memory_to_sink"] = {"what": "bsk_pipeline_open_memory + bsk_pipeline_next / _release: chunks delivered to the caller in record order",
                                      "tuples": sink_run(_L.SINK_TUPLES, 1, n_mem_streams)}
        if kind not in STREAM:
            # (with ~25 B per read going down the workers' host-side 2-bit packing binds: twelve workers on the box's 16 CPUs -- 57 / 67 / 70
            # Gbases/s with 8 / 12 / 14 at scale 100, profiles/NOTEBOOK.md 5.3)
            n_sets_streams = int(os.environ.get("BSK_BENCH_SETS_STREAMS", str(max(n_mem_streams, 12))))
            out["from_memory_to_sink"]["sets_scale_1"] = sink_run(_L.SINK_SETS, 1, n_sets_streams)
            out["from_memory_to_sink"]["sets_scale_100"] = sink_run(_L.SINK_SETS, 100, n_sets_streams)
            out["from_memory_to_sink"]["sets_note"] = ("BSK_SINK_SETS: per read the ascending distinct hashes with hash <= MaxUint64 / scale (iterator.go:181-185), reduced on "
                                                       "the device; scale 1 still moves ~8 B per distinct value, scale 100 moves what a FracMinHash consumer keeps")
    except Exception as e:  # the line must not be lost to this section
        out["from_memory_to_sink"] = {"error": repr(e)}
    # the same reads as files: fixed-width names, constant qualities
    rec = 12 + read_len + (3 + read_len if not alpha else 0)
    import shutil
    free = shutil.disk_usage(tempfile.gettempdir()).free
    n_file = max(1000, min(n_mem, int(0.25 * free / (rec + 1))))  # the sample file never takes more than a quarter of the free space
    arr = np.empty((n_file, rec), np.uint8)
    names = np.char.zfill(np.arange(n_file).astype("U9"), 9)
    arr[:, 0] = ord(">") if alpha else ord("@")
    arr[:, 1] = ord("r")
    arr[:, 2:11] = np.frombuffer("".join(names.tolist()).encode(), np.uint8).reshape(n_file, 9)
    arr[:, 11] = 10
    arr[:, 12:12 + read_len] = data[: n_file * read_len].reshape(n_file, read_len)
    if not alpha:
        arr[:, 12 + read_len] = 10
        arr[:, 13 + read_len] = ord("+")
        arr[:, 14 + read_len] = 10
        arr[:, 15 + read_len:15 + 2 * read_len] = ord("I")
        arr = np.concatenate([arr, np.full((n_file, 1), 10, np.uint8)], axis=1)
    else:
        arr = np.concatenate([arr, np.full((n_file, 1), 10, np.uint8)], axis=1)
    with tempfile.TemporaryDirectory() as td:
        plain = os.path.join(td, "reads.fx")
        arr.tofile(plain)
        n_gz = min(n_file, int(500_000 * scale))
        gzp = os.path.join(td, "reads.fx.gz")
        with open(gzp, "wb") as f:
            f.write(gzip.compress(arr[:n_gz].tobytes(), 1))
        for tag, path in (("from_plain_file", plain), ("from_gzip_file", gzp)):
            st = S.Engine.pipeline_fastx(path, p, n_streams=n_streams, chunk_records=1 << 18, fetch=True, alphabet=alpha)
            out[tag] = {"value": round(st["bases"] / st["seconds"] / 1e9, 3), "unit": "Gresidues/s" if alpha else "Gbases/s",
                        "reads": st["records"], "file_bytes": os.path.getsize(path), "seconds": round(st["seconds"], 4),
                        "reader_seconds": round(st["reader_seconds"], 4), "reader_wait_seconds": round(st["reader_wait_seconds"], 4),
                        "reader": ("block-parallel, %d parser threads, %d pieces re-parsed" % (st["reader_threads"], st["reparsed_pieces"]))
                        if st["reader_threads"] else "serial record reader (one gzip stream)",
                        "bound": "the record reader" if st["reader_seconds"] > 0.7 * st["seconds"] else "device side"}
        # BGZF (bgzip): gzip members of <= 64 KiB that record their own size -- located without inflating, inflated by the parser threads
        import struct
        import zlib
        n_bz = min(n_file, 4 * n_gz)
        raw = arr[:n_bz].tobytes()
        bz = bytearray()
        for i in list(range(0, len(raw), 0xff00)) + [len(raw)]:
            c = raw[i:i + 0xff00] if i < len(raw) else b""
            co = zlib.compressobj(1, zlib.DEFLATED, -15)
            comp = co.compress(c) + co.flush()
            bz += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(comp) + 8 - 1)
            bz += comp + struct.pack("<II", zlib.crc32(c) & 0xffffffff, len(c))
        bzp = os.path.join(td, "reads.bgzf.gz")
        with open(bzp, "wb") as f:
            f.write(bz)
        st = S.Engine.pipeline_fastx(bzp, p, n_streams=n_streams, chunk_records=1 << 18, fetch=True, alphabet=alpha)
        out["from_bgzf_file"] = {"value": round(st["bases"] / st["seconds"] / 1e9, 3), "unit": "Gresidues/s" if alpha else "Gbases/s",
                                 "reads": st["records"], "file_bytes": len(bz), "seconds": round(st["seconds"], 4),
                                 "reader": "block-parallel over the uncompressed text, %d threads inflate + parse" % st["reader_threads"]}
        del raw, bz
        # gzip scales by files, not inside one: eight gzip files of n_gz reads each, read at once (bsk_pipeline_fastx_files)
        gzs = []
        for i in range(8):
            gp = os.path.join(td, "part%d.fx.gz" % i)
            with open(gp, "wb") as f:
                f.write(gzip.compress(arr[i * n_gz:(i + 1) * n_gz].tobytes(), 1))
            gzs.append(gp)
        st = S.Engine.pipeline_fastx_files(gzs, p, n_streams=n_streams, n_readers=8, chunk_records=1 << 18, fetch=True, alphabet=alpha)
        out["from_8_gzip_files"] = {"value": round(st["bases"] / st["seconds"] / 1e9, 3), "unit": "Gresidues/s" if alpha else "Gbases/s",
                                    "reads": st["records"], "seconds": round(st["seconds"], 4), "readers": 8,
                                    "reader": "one serial record reader (zlib stream) per file, eight files at once"}
    S.Engine.pipeline_trim()
    return out


def measured_profile(workload: str, n_reads: int):
    """HBM bytes per launch / VALU utilisation from the committed rocprofv3 PMC passes (profiles/*/traffic.json), or None.

    bench.py cannot run the profiler on itself; the counters were collected on this same command in separate --pmc passes
    (FETCH_SIZE, WRITE_SIZE; SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES) and corrected as MI355X_MICROARCH.md prescribes.
    The entry names the kernel it was measured on: a stale file shows as a kernel-name mismatch in the JSON line.
    """
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for d in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        f = os.path.join(pdir, d, "traffic.json")
        if os.path.exists(f):
            for e in json.load(open(f)).get("entries", []):
                if e.get("workload") == workload and e.get("reads_per_gpu") == n_reads:
                    best = dict(e, profile=d)
    return best


def same_kernel(profiled: str, planned: str) -> bool:
    """The profiled instantiation (rocprofv3's full name, e.g. k_minimizer_ring<11,3,true>) against the plan's name (the planner prints the
    arguments a reader needs: k_minimizer_ring<11,true>): same family, and the plan's arguments appear in order among the profiled ones."""
    import re
    planned = planned.split(" ")[0]
    if profiled.split("<")[0] != planned.split("<")[0]:
        return False
    pa = re.findall(r"[\w-]+", profiled.partition("<")[2])
    it = iter(pa)
    return all(any(a == b for b in it) for a in re.findall(r"[\w-]+", planned.partition("<")[2]))


def plumbing_counters(rank: int):
    """Stand-in counters of --plumbing-only: [nanoseconds, bases, tuples, first-window-tie reads, non-ACGT reads]."""
    return [int((1.0 + 0.5 * rank) * 1e9), 15_000_000_000 + rank, 2_211_224_063 + 7 * rank, 88 + rank, 0]


def power_probe(eng, batch, p, res, k_ms, device, bases_per_launch, seconds=2.5):
    """Board power and shader clock while the SAME launch loops for a few seconds (outside the timed region): rocm-smi samples from a
    side thread.  The sketch kernels run against the board's power limit, not against a pipe: this is the evidence."""
    import re
    import shutil
    import subprocess
    import threading
    smi = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(smi):
        return None
    samples, stop = [], threading.Event()

    def sample():
        while not stop.is_set():
            try:
                o = subprocess.run([smi, "-d", str(device), "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
                pw = re.search(r"Package Power \(W\):\s*([0-9.]+)", o)
                sc = re.search(r"sclk clock level:\s*\S+\s*\((\d+)Mhz\)", o)
                if pw and sc:
                    samples.append((float(pw.group(1)), int(sc.group(1))))
            except Exception:
                pass
            stop.wait(0.15)

    cap = None
    try:
        o = subprocess.run([smi, "-d", str(device), "--showmaxpower"], capture_output=True, text=True, timeout=5).stdout
        m = re.search(r"Max Graphics Package Power \(W\):\s*([0-9.]+)", o)
        cap = float(m.group(1)) if m else None
    except Exception:
        pass
    iters = max(8, int(seconds * 1e3 / max(k_ms, 0.05)))
    eng.run_timed(batch, p, 0, max(4, iters // 5), reuse=res)  # the board warms up before the first sample
    th = threading.Thread(target=sample, daemon=True)
    th.start()
    _, ms = eng.run_timed(batch, p, 0, iters, reuse=res)
    stop.set()
    th.join(timeout=6)
    busy = [x for x in samples if x[1] > 500]  # (a sample taken before / after the loop shows the idle clock)
    if not busy:
        return None
    pw = statistics.median(x[0] for x in busy)
    rate = bases_per_launch / (sum(ms) / len(ms) * 1e-3)
    return {"board_power_w": pw, "power_cap_w": cap, "sclk_mhz": statistics.median(x[1] for x in busy), "samples": len(busy),
            "loop_seconds": round(sum(ms) / 1e3, 2), "units_per_s_in_loop": round(rate / 1e9, 1), "nj_per_unit": round(pw / rate * 1e9, 4),
            "note": "rocm-smi samples while the same launch loops (untimed, after the measurement); nj_per_unit = board power / bases (residues) per second"}


def self_launch(n: int) -> int:
    """--gpus N > 1 without a launcher: start N ranks of this very command line through torch.distributed.run (one process per GPU,
    rendezvous on 127.0.0.1 at a free port) and hand its exit code on; rank 0's JSON line goes to our stdout untouched."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def git_commit():
    """HEAD of the tree bench.py runs from (None outside a git checkout: the GPU box gets a snapshot without .git, so the round's
    profile script passes it in BSK_BENCH_COMMIT)."""
    c = os.environ.get("BSK_BENCH_COMMIT")
    if c:
        return c
    try:
        import subprocess
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="minimizer", choices=sorted(WORKLOADS))
    ap.add_argument("--reads", type=float, default=0, help="override reads per GPU (dev)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the file/host-memory -> host-tuples side measurement")
    ap.add_argument("--no-power-probe", action="store_true", help="skip the rocm-smi power / clock samples taken while the launch loops after the measurement")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend of the rendezvous / barriers")
    ap.add_argument("--plumbing-only", action="store_true", help="tests: N>1 control flow with stand-in counters, no GPU, no kernel")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not launched and (args.gpus > 1 or os.environ.get("BSK_BENCH_SELF_LAUNCH")):  # (BSK_BENCH_SELF_LAUNCH: tests take the self-launched form on a one-GPU box)
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:  # (a line that says n_gpus: 1 for --gpus 8 would be a measurement of the wrong thing)
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch {args.gpus} ranks (or drop the launcher environment and let bench.py start them)")
    if args.plumbing_only and args.backend != "gloo":
        raise SystemExit("--plumbing-only is the CPU test mode: use --backend gloo")

    import torch
    import torch.distributed as dist

    dev = None
    if not args.plumbing_only:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the sketch engine has no CPU fallback")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    # BSK_BENCH_FORCE_COMM=1: take the multi-rank path -- torch `nccl` process group alive, bsk_comm_unique_id -> bsk_comm_init_rank ->
    # bsk_gather_counts -- at world size 1 too (tests/test_gpu_comm.py: the only way to run that path on a one-GPU box)
    force_comm = bool(os.environ.get("BSK_BENCH_FORCE_COMM")) and not args.plumbing_only
    if world > 1 or force_comm:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier():
        if world > 1 or force_comm:
            dist.barrier()
        if dev is not None:
            torch.cuda.synchronize()

    kind, n_reads, read_len, k, x, desc = WORKLOADS[args.workload]
    if args.reads:
        n_reads = int(args.reads)

    eng = batch = res = None
    kernel_ms = []
    plan = {"kernel": "none (plumbing-only)", "grid": 0, "waves_per_cu": 0}
    gather_via = "bsk_gather_counts (RCCL)"
    if args.plumbing_only:
        barrier()
        barrier()
        mine = plumbing_counters(rank)
    else:
        from bio_amd import _lib as L
        from bio_amd import sketches as S

        seed = 0x5EED0000 + 3 + 0x1000000 * rank  # each rank hashes its own shard of the synthetic stream
        t_fc = time.perf_counter()
        eng = S.Engine(local_rank)
        first_call = {"bsk_ctx_create_ms": round((time.perf_counter() - t_fc) * 1e3, 1)}  # (what a real caller pays once per context: reported, never inside `value`)
        gather_via = "bsk_gather_counts (RCCL)"
        if world > 1 or force_comm:  # the communicator of the one collective: RCCL behind the C ABI; the id travels over the launcher's store
            try:
                uid = [eng.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                eng.comm_init_rank(uid[0], rank, world)
            except Exception as e:  # never lose the run to plumbing: fall back to the launcher's own collective, and say so
                gather_via = f"torch.distributed all_gather (bsk_comm_init_rank failed: {e!r})"
            # every rank must take the same path
            flag = torch.tensor([0 if gather_via.startswith("bsk") else 1], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()) and gather_via.startswith("bsk"):
                gather_via = "torch.distributed all_gather (another rank could not join the bsk communicator)"
        batch = eng.synth(L.ALPHA_PROTEIN if kind in PROTEIN else L.ALPHA_DNA, n_reads, read_len, seed)
        p = {"min": lambda: eng.params(L.MINIMIZER, k, w=x), "nt": lambda: eng.params(L.NTHASH, k), "syn": lambda: eng.params(L.SYNCMER, k, s=x),
             "pmin": lambda: eng.params(L.PROT_MINIMIZER, k, w=x), "kmer": lambda: eng.params(L.KMER, k),
             "phash": lambda: eng.params(L.PROT_HASH, k), "sim": lambda: eng.params(L.SIMHASH, k, m=x, scale=5)}[kind]()
        # untimed: first run sizes the result buffers; then W warm-up steps
        t_fc = time.perf_counter()
        res = eng.run(batch, p)
        first_call["first_bsk_sketch_ms"] = round((time.perf_counter() - t_fc) * 1e3, 1)  # (plans, allocates the result arrays, one sizing launch)
        res, _ = eng.run_timed(batch, p, args.warmup, 0, reuse=res)
        barrier()
        t0 = time.perf_counter()
        res, kernel_ms = eng.run_timed(batch, p, 0, args.steps, reuse=res)  # exactly K steps, HIP events around each kernel
        barrier()
        dt = time.perf_counter() - t0
        dg = res.digest()
        plan = res.plan()
        mine = [int(dt * 1e9), n_reads * read_len, int(res.info()["n_tuples"]), int(dg["first_window_tie"]), int(dg["has_non_acgt"])]

    # whole-job numbers: MAX time over ranks, SUM of units over ranks -- ONE all_gather of five u64 counters per rank
    if world == 1 and not force_comm:
        rows = [mine]
    elif args.plumbing_only:
        t = torch.tensor(mine, dtype=torch.int64)
        outl = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(outl, t)
        rows = [[int(v) for v in o.tolist()] for o in outl]
    elif gather_via.startswith("bsk"):
        rows = eng.gather_counts(mine)  # bsk_gather_counts: RCCL all_gather
    else:
        t = torch.tensor(mine, dtype=torch.int64, device=dev)
        outl = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(outl, t)
        rows = [[int(v) for v in o.tolist()] for o in outl]
    from bio_amd.shard import whole_job
    job = whole_job([[r[0] / 1e9, float(r[1]), float(r[2])] for r in rows], args.steps)
    dt_max, bases_total, tuples_total = job["seconds"], job["bases"], job["tuples"]
    tuples = mine[2]

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
            "dtype": "u64", "data": "synthetic" if not args.plumbing_only else "none (plumbing-only: stand-in counters, NO kernel ran; not a measurement)",
            "config": {"workload": desc, "reads_per_gpu": n_reads, "read_len": read_len, "k": k, par[0]: par[1],
                       "tuples_per_gpu": int(tuples), "tuples_total": int(tuples_total),
                       "first_window_tie_reads": int(sum(r[3] for r in rows)), "non_acgt_reads": int(sum(r[4] for r in rows)),
                       "per_rank_seconds": [round(r[0] / 1e9, 6) for r in rows],
                       "parallelism": f"reads sharded by record over {world} GPU(s), no data-path collective; counters gathered by "
                                      + (gather_via if (world > 1 or force_comm) and not args.plumbing_only else "torch gloo (plumbing-only)" if world > 1 else "nothing (1 GPU)"),
                       "input": ("residues (1 B each)" if kind in PROTEIN else "2-bit packed reads") + " resident in HBM",
                       "output": ("hash u64 per position" if kind in STREAM else "hash u64 + pos|strand u32") + " + u64 index per read, in HBM"},
        }
        if kernel_ms:
            k_ms = sum(kernel_ms) / len(kernel_ms)
            k_med = statistics.median(kernel_ms)
            achieved = alg_bytes / (k_ms * 1e-3) / 1e9
            prof = measured_profile(args.workload, n_reads)
            kern = plan["kernel"]
            out["roofline"] = {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": prof.get("hbm_bytes_per_launch") if prof else None,
                "kernel": kern, "grid": plan["grid"], "waves_per_cu": plan["waves_per_cu"],
                "kernel_ms_avg": round(k_ms, 4), "kernel_ms_median": round(k_med, 4), "kernel_ms_min": round(min(kernel_ms), 4),
                "launches_timed": len(kernel_ms),
                "algorithmic_bytes_per_launch": int(alg_bytes),
                "frac_of_6.29TBps_copy_ceiling": round(achieved / HBM_COPY_GBS, 4),
                "read_only_frac": round(in_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "valu": prof.get("valu") if prof else None,
                "binding_ceiling": (("valu-issue" if (prof.get("valu") or {}).get("frac", 0) > achieved / HBM_PEAK_GBS else "hbm") if prof and prof.get("valu") else None),
                "profile": ({"dir": "profiles/" + prof["profile"], "kernel": prof.get("kernel"), "commit": prof.get("commit"), "this_run_commit": git_commit(),
                             "kernel_matches_this_run": same_kernel(prof.get("kernel", ""), kern)}
                            if prof else None),
                "note": NOTES[kind] + "; frac is vs the 8 TB/s spec peak; read_only_frac = input bytes alone over the same peak "
                                      "(north_star's 'HBM-read roofline'); `bound`/`achieved`/`peak` are the HBM roofline of the contract; `valu` is the VALU-issue "
                                      "roofline of the committed PMC pass (SQ_INSTS_VALU x the mean issue cost of the kernel's inner-loop instruction mix, "
                                      "scripts/valu_model.py, over the dispatch's SIMD-cycles) and `binding_ceiling` names the larger of the two fractions",
            }
        if kernel_ms and not args.no_power_probe and not args.plumbing_only:
            try:
                out["roofline"]["power"] = power_probe(eng, batch, p, res, sum(kernel_ms) / len(kernel_ms), local_rank, n_reads * read_len)
            except Exception as e:
                out["roofline"]["power"] = {"error": repr(e)}
        if not args.plumbing_only:
            out["first_call"] = dict(first_call, note="one-time costs of a context / a fresh result on this rank (wall ms): outside the timed region, reported so that a caller can see them")
        if world == 1 and kernel_ms and not args.plumbing_only:
            # what a caller that releases every result and sketches the next batch pays per call: bsk_sketch's wall time on a FRESH result
            # (plan + result arrays + one sizing launch + the totals read back) -- outside `value`, reported beside it.  The first of the
            # three allocates its arrays (the timed result still holds its own); the next ones take over what the one before released.
            try:
                pc = []
                for _ in range(3):
                    t_pc = time.perf_counter()
                    r2 = eng.run(batch, p)
                    pc.append(round((time.perf_counter() - t_pc) * 1e3, 3))
                    r2.close()
                out["plain_call"] = {"ms": pc, "steady_ms": min(pc[1:]), "kernel_ms_avg": round(sum(kernel_ms) / len(kernel_ms), 4),
                                     "note": "wall ms of bsk_sketch on a fresh result, three calls in a row, each result released before the next call "
                                             "(the context keeps a released result's arrays for the next one: bsk_ctx::spare, NOTEBOOK 6.7)"}
            except Exception as e:
                out["plain_call"] = {"error": repr(e)}
        if world == 1 and kernel_ms and not args.plumbing_only:
            # what a device-side consumer of the tuples pays on top of the sketch: bsk_result_compact (offsets scanned, the units'
            # slabs squeezed into dense CSR arrays left in HBM; sets.hip, k_gather_groups) -- outside `value`, reported beside it
            try:
                ts = []
                for _ in range(4):
                    t0 = time.perf_counter()
                    _, _, _, nt = res.compact()
                    ts.append((time.perf_counter() - t0) * 1e3)
                c_ms = min(ts[1:])
                per = 12 if kind not in STREAM else 8
                moved = nt * 2 * per + n_reads * 24
                out["dense_copy"] = {"what": "bsk_result_compact: dense CSR copy of the result left on the device (wall time of the call, incl. its two synchronisations; "
                                             "first call, which allocates the arrays, not counted)",
                                     "ms": round(c_ms, 3), "first_call_ms": round(ts[0], 3), "tuples": int(nt), "bytes_moved": int(moved),
                                     "GB_per_s": round(moved / c_ms / 1e6, 1),
                                     "sketch_plus_dense_copy": round(n_reads * read_len / ((sum(kernel_ms) / len(kernel_ms)) + c_ms) / 1e6, 2), "unit": out["unit"]}
            except Exception as e:
                out["dense_copy"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline and not args.plumbing_only:
            out["cpu_baseline"] = cpu_baseline(kind, k, x, read_len, batch)
        if world == 1 and not args.no_end_to_end and not args.plumbing_only:
            try:
                out["end_to_end"] = end_to_end(kind, p, batch, read_len, int(os.environ.get("BSK_BENCH_STREAMS", "5")))
            except Exception as e:  # never let the side measurement cost the line
                out["end_to_end"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1 or force_comm:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
