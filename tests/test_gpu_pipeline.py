"""The end-to-end path (bsk_pipeline_fastx / bsk_pipeline_memory / bsk_batch_refill_ascii): file or host memory -> pinned chunks ->
H2D + pack -> kernel -> tuples on the host over several streams must give exactly what one big batch gives."""
import ctypes as C
import gzip
import os
import random

import numpy as np
import pytest

from bio_amd import _lib as L
from bio_amd import sketches as S

pytestmark = pytest.mark.gpu


def make_reads(n, seed, with_n=True):
    rng = np.random.default_rng(seed)
    lens = rng.integers(30, 260, n)
    lens[::7] = 150
    offs = np.zeros(n + 1, np.uint64)
    offs[1:] = np.cumsum(lens)
    data = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(offs[-1]))].copy()
    if with_n:
        data[rng.integers(0, len(data), n // 50)] = ord("N")  # ~2 % of the reads take the ASCII side launch
    return data, offs


def write_fastq(path, data, offs, gz):
    opener = gzip.open if gz else open
    with opener(path, "wb") as f:
        for i in range(len(offs) - 1):
            s = data[int(offs[i]):int(offs[i + 1])].tobytes()
            f.write(b"@r%d some description\n%s\n+\n%s\n" % (i, s, b"I" * len(s)))


@pytest.mark.parametrize("kind,pk", [(L.MINIMIZER, dict(k=21, w=11)), (L.SYNCMER, dict(k=31, s=11)), (L.NTHASH, dict(k=21))])
def test_pipeline_equals_one_batch(engine, tmp_path, kind, pk):
    n = 60_000
    data, offs = make_reads(n, 3)
    p = engine.params(kind, **pk)
    whole = engine.run(engine.batch_from_arrays(data, offs), p)
    want = whole.digest()
    for streams, chunk, fetch in ((1, 7000, True), (3, 5001, True), (2, 100000, False)):
        st = S.Engine.pipeline_memory(data, offs, p, n_streams=streams, chunk_records=chunk, fetch=fetch)
        assert st["records"] == n and st["bases"] == int(offs[-1]) and st["tuples"] == want["n_tuples"], (streams, st)
        assert st["checksum"] == want["checksum"], (streams, chunk, fetch)
        assert st["chunks"] == -(-n // chunk) and st["n_streams"] == streams and st["seconds"] > 0
    for gz in (False, True):
        path = str(tmp_path / ("reads.fq.gz" if gz else "reads.fq"))
        write_fastq(path, data, offs, gz)
        st = S.Engine.pipeline_fastx(path, p, n_streams=2, chunk_records=9000, fetch=True)
        assert (st["records"], st["bases"], st["tuples"], st["checksum"]) == (n, int(offs[-1]), want["n_tuples"], want["checksum"])
        assert (st["reader_threads"] > 0) == (not gz)  # plain files: the block-parallel reader; gzip: one serial stream
        if not gz:  # pieces far smaller than a chunk, and the serial reader on the same file
            for env in ({"BSK_FASTX_PIECE": "20000", "BSK_FASTX_THREADS": "5"}, {"BSK_FASTX_SERIAL": "1"}):
                os.environ.update(env)
                try:
                    st2 = S.Engine.pipeline_fastx(path, p, n_streams=3, chunk_records=7001, fetch=True)
                finally:
                    for k_ in env:
                        del os.environ[k_]
                assert (st2["records"], st2["tuples"], st2["checksum"]) == (n, want["n_tuples"], want["checksum"]), env
                assert (st2["reader_threads"] == 0) == ("BSK_FASTX_SERIAL" in env)
    st = S.Engine.pipeline_memory(data, offs, p, n_streams=2, chunk_records=20000, repeat=3)
    assert st["records"] == 3 * n and st["checksum"] == (3 * want["checksum"]) % (1 << 64)


def test_refill_reuses_one_batch_object(engine):
    """bsk_batch_refill_ascii: growing, shrinking, pure-ACGT and N-carrying chunks through ONE batch object, every result equal to
    a fresh batch's."""
    lib = engine.lib
    h = C.c_void_p()
    p = engine.params(L.MINIMIZER, 15, w=7)
    for i, (n, with_n) in enumerate([(500, False), (4000, True), (100, False), (4000, False), (1, True), (9000, True), (0, False)]):
        data, offs = make_reads(max(n, 1), 10 + i, with_n)
        if n == 0:
            data, offs = np.zeros(1, np.uint8), np.zeros(1, np.uint64)
        engine._chk(lib.bsk_batch_refill_ascii(engine.ctx, C.byref(h), data.ctypes.data, offs.ctypes.data, n, L.ALPHA_DNA))
        got = S.BatchResult(engine, *_run(engine, h, p)).digest()
        want = engine.run(engine.batch_from_arrays(data, offs) if n else engine.batch([]), p).digest()
        assert got == want, (i, n)
    lib.bsk_batch_destroy(h)


def _run(engine, batch_handle, p):
    r = C.c_void_p()
    engine._chk(engine.lib.bsk_sketch(engine.ctx, batch_handle, C.byref(p), C.byref(r)))
    return r, p


def test_pipeline_reports_reader_errors(tmp_path):
    path = str(tmp_path / "bad.fq")
    with open(path, "w") as f:
        f.write("@r1\nACGTACGTACGTACGTACGTACGTACGT\n+\nIIII\n")  # quality shorter than the sequence
    with pytest.raises(S.DeviceError):
        S.Engine.pipeline_fastx(path, S.Engine.params(L.NTHASH, 5), n_streams=2, chunk_records=10)


def test_pipeline_and_refill_protein(engine, tmp_path):
    """Protein FASTA through the pipeline (alphabet guessed from the first record, as the reference's reader does) and protein chunks
    through one re-filled batch object."""
    rng = random.Random(9)
    aa = "ACDEFGHIKLMNPQRSTVWY"
    seqs = ["".join(rng.choice(aa) for _ in range(rng.randint(5, 420))) for _ in range(5000)]
    path = str(tmp_path / "prot.fa")
    with open(path, "w") as f:
        for i, s in enumerate(seqs):
            f.write(">p%d\n" % i)
            for j in range(0, len(s), 60):  # multi-line FASTA
                f.write(s[j:j + 60] + "\n")
    p = engine.params(L.PROT_MINIMIZER, 9, w=5)
    want = engine.run(engine.batch(seqs, L.ALPHA_PROTEIN), p).digest()
    st = S.Engine.pipeline_fastx(path, p, n_streams=2, chunk_records=700, fetch=True)
    assert (st["records"], st["tuples"], st["checksum"]) == (5000, want["n_tuples"], want["checksum"])
    h = C.c_void_p()
    for lo, hi in ((0, 900), (900, 1000), (1000, 5000)):
        chunk = seqs[lo:hi]
        data = np.frombuffer("".join(chunk).encode(), np.uint8)
        offs = np.zeros(len(chunk) + 1, np.uint64)
        offs[1:] = np.cumsum([len(s) for s in chunk])
        engine._chk(engine.lib.bsk_batch_refill_ascii(engine.ctx, C.byref(h), data.ctypes.data, offs.ctypes.data, len(chunk), L.ALPHA_PROTEIN))
        got = S.BatchResult(engine, *_run(engine, h, p)).digest()
        assert got == engine.run(engine.batch(chunk, L.ALPHA_PROTEIN), p).digest(), (lo, hi)
    engine.lib.bsk_batch_destroy(h)


def test_pipeline_reports_a_damaged_file(engine, tmp_path):
    """A record with a longer quality than sequence (ErrBadFASTQFormat, reader.go:19) in the middle of a plain FASTQ: both readers stop
    the pipeline with the reader's error -- and the run leaves nothing behind (a second run on a good file works)."""
    data, offs = make_reads(30_000, 9, with_n=False)
    good = str(tmp_path / "good.fq")
    write_fastq(good, data, offs, False)
    text = open(good, "rb").read()
    cut = text.index(b"@r15000 ")
    bad = str(tmp_path / "bad.fq")
    with open(bad, "wb") as f:
        f.write(text[:cut] + b"@broken\nACGT\n+\nIIIIIIII\n" + text[cut:])
    p = engine.params(L.MINIMIZER, 21, w=11)
    for env in ({"BSK_FASTX_PIECE": "100000"}, {"BSK_FASTX_SERIAL": "1"}):
        os.environ.update(env)
        try:
            with pytest.raises(S.DeviceError, match="unknown|bad|fastq|71|BSK|bsk"):
                S.Engine.pipeline_fastx(bad, p, n_streams=2, chunk_records=4000, fetch=True)
            st = S.Engine.pipeline_fastx(good, p, n_streams=2, chunk_records=4000, fetch=True)
            assert st["records"] == 30_000
        finally:
            for k_ in env:
                del os.environ[k_]


def test_pipeline_over_several_files(engine, tmp_path):
    """bsk_pipeline_fastx_files: plain and gzip files, FASTQ and FASTA, read concurrently by several producers: the job's records,
    tuples and checksum are the sums over the files (each file == one batch of its reads)."""
    p = engine.params(L.MINIMIZER, 21, w=11)
    paths, want = [], dict(records=0, tuples=0, checksum=0)
    for i in range(7):
        data, offs = make_reads(4000 + 3000 * i, 30 + i)
        d = engine.run(engine.batch_from_arrays(data, offs), p).digest()
        want["records"] += len(offs) - 1
        want["tuples"] += d["n_tuples"]
        want["checksum"] = (want["checksum"] + d["checksum"]) % (1 << 64)
        path = str(tmp_path / ("f%d.fq%s" % (i, ".gz" if i % 2 else "")))
        if i == 4:  # one FASTA among them
            path = str(tmp_path / "f4.fa")
            with open(path, "wb") as f:
                for j in range(len(offs) - 1):
                    f.write(b">s%d\n%s\n" % (j, data[int(offs[j]):int(offs[j + 1])].tobytes()))
        else:
            write_fastq(path, data, offs, bool(i % 2))
        paths.append(path)
    for readers, streams, chunk in ((0, 2, 3000), (1, 1, 100000), (3, 3, 1777), (7, 2, 5000)):
        st = S.Engine.pipeline_fastx_files(paths, p, n_streams=streams, n_readers=readers, chunk_records=chunk)
        assert (st["records"], st["tuples"], st["checksum"]) == (want["records"], want["tuples"], want["checksum"]), (readers, streams, chunk)
    with pytest.raises(S.DeviceError):
        S.Engine.pipeline_fastx_files(paths + [str(tmp_path / "missing.fq")], p)


def test_pinned_buffers_are_pooled_between_runs(engine):
    """A finished run parks its pinned buffers in the process-wide pool: the next run of the same shape pins (almost) nothing,
    and bsk_pipeline_trim() empties the pool again.  Results do not depend on where the buffers came from."""
    n = 400_000
    data, offs = make_reads(n, 11, with_n=False)
    p = engine.params(L.MINIMIZER, k=21, w=11)
    S.Engine.pipeline_trim()
    first = S.Engine.pipeline_memory(data, offs, p, n_streams=2, chunk_records=100_000, fetch=True)
    again = S.Engine.pipeline_memory(data, offs, p, n_streams=2, chunk_records=100_000, fetch=True)
    S.Engine.pipeline_trim()
    fresh = S.Engine.pipeline_memory(data, offs, p, n_streams=2, chunk_records=100_000, fetch=True)
    S.Engine.pipeline_trim()
    assert first["checksum"] == again["checksum"] == fresh["checksum"] and first["tuples"] == again["tuples"] == fresh["tuples"]
    assert again["pin_seconds"] < 0.5 * first["pin_seconds"], (first["pin_seconds"], again["pin_seconds"])
    assert fresh["pin_seconds"] > again["pin_seconds"], (fresh["pin_seconds"], again["pin_seconds"])


def test_narrow_fetch_equals_wide_fetch(engine):
    """bsk_result_fetch_narrow (u32 offsets scanned on the device, u16 positions) against bsk_result_fetch: whole results and ranges,
    slab (k_minimizer_pk), unit-row (k_minimizer_ring) and per-read-slab layouts, a stream kind; a read too long for 15-bit positions."""
    import random
    from bio_amd import _lib as L
    from bio_amd import sketches as S
    rng = random.Random(4)
    for rl, k, w in ((150, 21, 11), (250, 21, 11), (150, 15, 3)):
        b = engine.synth(L.ALPHA_DNA, 20_000, rl, 0x5EED0003)
        res = engine.run(b, engine.params(L.MINIMIZER, k, w=w))
        for first, count in ((0, None), (5_000, 7_777), (19_999, 1), (3, 0)):
            o, st, h, p = res.fetch(first, count)
            o2, st2, h2, p2 = res.fetch_narrow(first, count)
            assert np.array_equal(o, o2.astype(np.uint64)) and np.array_equal(st, st2) and np.array_equal(h, h2)
            assert np.array_equal(p & L.POS_MASK, (p2 & 0x7FFF).astype(np.uint32)) and np.array_equal(p >> 31, (p2 >> 15).astype(np.uint32))
        res.close()
        b.close()
    b = engine.synth(L.ALPHA_DNA, 3_000, 150, 0x5EED0003)
    res = engine.run(b, engine.params(L.NTHASH, 21))
    o, st, h, p = res.fetch()
    o2, st2, h2, p2 = res.fetch_narrow()
    assert p is None and p2 is None and np.array_equal(o, o2.astype(np.uint64)) and np.array_equal(h, h2)
    res.close()
    b.close()
    long_read = "".join(rng.choice("ACGT") for _ in range(40_000))
    b = engine.batch([long_read, long_read[:100]])
    res = engine.run(b, engine.params(L.MINIMIZER, 21, w=11))
    with pytest.raises(S.DeviceError):
        res.fetch_narrow()
    res.close()
    b.close()


def test_host_packed_chunks_equal_ascii_chunks(engine, monkeypatch):
    """bsk_pipeline_memory packs chunks of pure ACGT reads to 2-bit words on its worker threads (bsk_batch_refill_packed) and sends the
    others as ASCII: mixed case, chunks with and without an N, reads whose length is not a multiple of 16 or 32, the last read of the
    source (no room for a whole vector block), zero-length reads -- all against one batch, and against BSK_PIPE_NO_HOST_PACK."""
    rng = np.random.default_rng(21)
    n = 50_000
    lens = rng.integers(0, 200, n).astype(np.uint64)
    lens[::5] = 150
    lens[-1] = 37
    offs = np.zeros(n + 1, np.uint64)
    offs[1:] = np.cumsum(lens)
    data = np.frombuffer(b"ACGTacgt", np.uint8)[rng.integers(0, 8, int(offs[-1]))].copy()
    for with_n in (False, True):
        d = data.copy()
        if with_n:
            d[int(offs[12_345]) + 3] = ord("N")      # one chunk of several goes the ASCII way
            d[int(offs[40_000]) + 1] = ord("-")
        p = engine.params(L.MINIMIZER, 21, w=11)
        want = engine.run(engine.batch_from_arrays(d, offs), p).digest()
        for fetch in (True, False):
            st = S.Engine.pipeline_memory(d, offs, p, n_streams=3, chunk_records=6000, fetch=fetch)
            assert (st["records"], st["tuples"], st["checksum"]) == (n, want["n_tuples"], want["checksum"]), (with_n, fetch)
        monkeypatch.setenv("BSK_PIPE_NO_HOST_PACK", "1")
        st2 = S.Engine.pipeline_memory(d, offs, p, n_streams=2, chunk_records=6000, fetch=True)
        monkeypatch.delenv("BSK_PIPE_NO_HOST_PACK")
        assert (st2["tuples"], st2["checksum"]) == (want["n_tuples"], want["checksum"])
    # refill through ONE batch object, packed and ASCII chunks in turn
    lib, h = engine.lib, C.c_void_p()
    p = engine.params(L.MINIMIZER, 15, w=7)
    for i, packed in enumerate([True, False, True, True, False]):
        m = [3000, 500, 7000, 10, 4000][i]
        dd, oo = data[: int(offs[m])], offs[: m + 1]
        if packed:
            words, desc = [], np.zeros(m, np.uint64)
            w = 0
            for r in range(m):
                s = dd[int(oo[r]):int(oo[r + 1])]
                codes = (((s >> 1) ^ (s >> 2)) & 3).astype(np.uint64)
                nw = (len(s) + 15) // 16
                pad = np.zeros(nw * 16, np.uint64)
                pad[: len(s)] = codes
                words.append((pad.reshape(nw, 16) << (2 * np.arange(16, dtype=np.uint64))).sum(axis=1).astype(np.uint32))
                desc[r] = (w << 24) | len(s)
                w += nw
            words = np.concatenate(words) if words else np.zeros(1, np.uint32)
            engine._chk(lib.bsk_batch_refill_packed(engine.ctx, C.byref(h), words.ctypes.data, w, desc.ctypes.data, m))
        else:
            engine._chk(lib.bsk_batch_refill_ascii(engine.ctx, C.byref(h), dd.ctypes.data, oo.ctypes.data, m, L.ALPHA_DNA))
        got = S.BatchResult(engine, *_run(engine, h, p)).digest()
        assert got == engine.run(engine.batch_from_arrays(dd, oo), p).digest(), (i, packed)
    lib.bsk_batch_destroy(h)


# ---- the pipeline with a consumer (bsk_pipeline_open_* / _next / _release / _close) ---------------------------------------------------
def _all_devices():
    n = C.c_int()
    L.load().bsk_device_count(C.byref(n))
    return list(range(max(1, n.value)))


def _collect(pl):
    """every chunk of the run, copied, in delivery order"""
    out = []
    for c in pl.chunks():
        out.append(dict(seq=c.sequence, first=c.first_record, n=c.n_records, src=c.source_index, offsets=None if c.offsets is None else c.offsets.astype(np.uint64).copy(), status=None if c.status is None else c.status.copy(),
                        hash=None if c.hash is None else c.hash.copy(), pos=None if c.pos is None else c.pos.copy(),
                        strand=None if c.strand is None else c.strand.copy(), link=c.link_bytes, n_tuples=c.n_tuples, n_values=c.n_values))
    return out


@pytest.mark.parametrize("devices", ["one", "twice", "all"])
@pytest.mark.parametrize("kind,pk", [(L.MINIMIZER, dict(k=21, w=11)), (L.SYNCMER, dict(k=31, s=11)), (L.NTHASH, dict(k=21))])
def test_sink_delivers_every_tuple_in_record_order(engine, devices, kind, pk):
    """What the sink hands out, chunk after chunk, is bsk_result_fetch of the same records of ONE big batch: same offsets, status bytes,
    hashes, positions and strands, in input order -- on one device, on one device named twice (the n > 1 code on a 1-GPU box) and on all."""
    devs = {"one": [0], "twice": [0, 0], "all": _all_devices()}[devices]
    n = 40_000
    data, offs = make_reads(n, 21)
    p = engine.params(kind, **pk)
    whole = engine.run(engine.batch_from_arrays(data, offs), p)
    w_off, w_st, w_h, w_p = whole.fetch()
    with S.Engine.pipeline_open(p, data=data, offsets=offs, devices=devs, n_streams=2, chunk_records=3001, sink=L.SINK_TUPLES, alphabet=L.ALPHA_DNA) as pl:
        chunks = _collect(pl)
    st = pl.stats
    assert [c["seq"] for c in chunks] == list(range(len(chunks))) and len(chunks) == -(-n // 3001)
    at = 0
    for c in chunks:
        assert c["first"] == at and c["src"] == 0
        m = c["n"]
        a, b = int(w_off[at]), int(w_off[at + m])
        assert np.array_equal(c["offsets"], w_off[at:at + m + 1] - w_off[at]), c["seq"]
        assert np.array_equal(c["status"], w_st[at:at + m])
        assert np.array_equal(c["hash"], w_h[a:b])
        if w_p is not None:
            assert np.array_equal(c["pos"], w_p[a:b] & L.POS_MASK) and np.array_equal(c["strand"], (w_p[a:b] >> 31).astype(np.uint8))
        else:
            assert c["pos"] is None
        at += m
    assert at == n and st["records"] == n and st["tuples"] == int(w_off[-1]) and st["n_streams"] == 2 * len(devs)


@pytest.mark.parametrize("scale", [1, 100])
def test_sink_sets_mode_equals_result_sets(engine, scale):
    """BSK_SINK_SETS: per record the ascending distinct values with hash <= MaxUint64 / scale (iterator.go:181-185) -- what bsk_result_sets
    gives for one big batch, and what numpy gives from the fetched tuples."""
    n = 30_000
    data, offs = make_reads(n, 22)
    p = engine.params(L.MINIMIZER, 21, w=11)
    whole = engine.run(engine.batch_from_arrays(data, offs), p)
    s_off, s_val = whole.sets(scale=scale)
    w_off, _, w_h, _ = whole.fetch()
    with S.Engine.pipeline_open(p, data=data, offsets=offs, devices=[0, 0], n_streams=2, chunk_records=4096, sink=L.SINK_SETS, sets_scale=scale, alphabet=L.ALPHA_DNA) as pl:
        chunks = _collect(pl)
    at = 0
    link = 0
    for c in chunks:
        m = c["n"]
        assert c["first"] == at
        assert np.array_equal(c["offsets"], s_off[at:at + m + 1] - s_off[at])
        assert np.array_equal(c["hash"], s_val[int(s_off[at]):int(s_off[at + m])])
        assert c["n_values"] == int(s_off[at + m] - s_off[at]) and c["pos"] is None
        link += c["link"]
        at += m
    assert at == n
    lim = np.uint64(0xFFFFFFFFFFFFFFFF // scale)
    for r in (0, 1, 77, n - 1):  # the definition itself, from the tuples
        v = np.unique(w_h[int(w_off[r]):int(w_off[r + 1])])
        assert np.array_equal(v[v <= lim], s_val[int(s_off[r]):int(s_off[r + 1])])
    assert link == 5 * n + 8 * int(s_off[-1])  # what crossed the link: u32 offsets + status bytes + the surviving values
    if scale == 100:  # the reduction is the point (window minima are small hashes: ~11 % of them pass hash <= max / 100, not 1 %)
        assert link < 0.25 * (10 * int(w_off[-1]) + 5 * n)


def test_sink_over_files_and_early_close(engine, tmp_path):
    """Files (plain: host-packed chunks of the block-parallel reader; gzip: the serial reader) through the sink; a consumer that stops early
    closes cleanly; the callback form (bsk_pipeline_run) sees the same chunks."""
    n = 25_000
    data, offs = make_reads(n, 23, with_n=False)
    p = engine.params(L.MINIMIZER, 21, w=11)
    whole = engine.run(engine.batch_from_arrays(data, offs), p)
    w_off, w_st, w_h, w_p = whole.fetch()
    paths = []
    for gz in (False, True):
        path = str(tmp_path / ("r.fq.gz" if gz else "r.fq"))
        write_fastq(path, data, offs, gz)
        paths.append(path)
    for path in paths:
        with S.Engine.pipeline_open(p, paths=[path], devices=[0], n_streams=3, chunk_records=2500, sink=L.SINK_TUPLES) as pl:
            chunks = _collect(pl)
        assert sum(c["n"] for c in chunks) == n
        assert np.array_equal(np.concatenate([c["hash"] for c in chunks]), w_h[: int(w_off[-1])])
        assert np.array_equal(np.concatenate([c["pos"] for c in chunks]), w_p[: int(w_off[-1])] & L.POS_MASK)
        assert [c["first"] for c in chunks] == list(np.cumsum([0] + [c["n"] for c in chunks[:-1]]))
    # two files, one reader: file 0's chunks, then file 1's, every file counted from its own record 0
    with S.Engine.pipeline_open(p, paths=paths, devices=[0], n_streams=2, chunk_records=6000, sink=L.SINK_COUNTS, n_readers=1) as pl:
        chunks = _collect(pl)
    assert [c["src"] for c in chunks] == sorted(c["src"] for c in chunks) and {c["src"] for c in chunks} == {0, 1}
    assert sum(c["n_tuples"] for c in chunks) == 2 * int(w_off[-1]) and all(c["hash"] is None for c in chunks)
    # early close
    pl = S.Engine.pipeline_open(p, data=data, offsets=offs, devices=[0], n_streams=2, chunk_records=1000, sink=L.SINK_TUPLES, alphabet=L.ALPHA_DNA, repeat=50)
    first = pl.next()
    assert first.sequence == 0 and first.n_records == 1000
    st = pl.close()
    assert st["seconds"] > 0
    # the callback form
    lib = L.load()
    seen = []

    def on_chunk(user, cptr):
        c = cptr.contents
        seen.append((c.sequence, c.first_record, c.n_records, c.n_tuples))
        return 0

    dev = (C.c_int * 1)(0)
    cfg = L.PipelineConfig(dev, 1, 2, 5000, L.SINK_TUPLES, 1, L.ALPHA_DNA, 0, 0, 0)
    h = C.c_void_p()
    d8, o64 = np.ascontiguousarray(data, np.uint8), np.ascontiguousarray(offs, np.uint64)
    assert lib.bsk_pipeline_open_memory(C.byref(cfg), d8.ctypes.data, o64.ctypes.data, n, 1, C.byref(p), C.byref(h)) == L.OK
    stats = L.PipelineStats()
    cb = L.CHUNK_FN(on_chunk)
    assert lib.bsk_pipeline_run(h, cb, None, C.byref(stats)) == L.OK
    assert [s[0] for s in seen] == list(range(5)) and sum(s[2] for s in seen) == n and sum(s[3] for s in seen) == int(w_off[-1]) == stats.tuples
    # a callback that stops the run: a code of its own, not an argument error
    seen.clear()
    stop_cb = L.CHUNK_FN(lambda user, cptr: 1)
    h = C.c_void_p()
    assert lib.bsk_pipeline_open_memory(C.byref(cfg), d8.ctypes.data, o64.ctypes.data, n, 20, C.byref(p), C.byref(h)) == L.OK
    assert lib.bsk_pipeline_run(h, stop_cb, None, C.byref(stats)) == L.ERR_STOPPED
    assert lib.bsk_err_name(L.ERR_STOPPED).decode().startswith("pipeline: stopped")


def test_mixed_plain_and_gzip_files_share_the_chunk_pool(engine, tmp_path):
    """Plain files are 2-bit packed on the host, gzip files arrive as ASCII from the serial reader, and both fill chunks from ONE free queue:
    a chunk that carried packed words must not keep that flag when the gzip reader fills it next (pipeline.cpp).  Several
    files read at once, tuples out, every file's chunks equal to its single-file run."""
    n = 12_000
    p = engine.params(L.MINIMIZER, 21, w=11)
    paths, want = [], []
    for i, gz in enumerate((False, True, False, True, True)):
        data, offs = make_reads(n + 500 * i, 31 + i, with_n=False)
        path = str(tmp_path / (f"m{i}.fq.gz" if gz else f"m{i}.fq"))
        write_fastq(path, data, offs, gz)
        paths.append(path)
        whole = engine.run(engine.batch_from_arrays(data, offs), p)
        w_off, _, w_h, w_p = whole.fetch()
        want.append((n + 500 * i, w_h[: int(w_off[-1])].copy(), (w_p[: int(w_off[-1])] & L.POS_MASK).copy()))
    for n_readers, n_streams in ((1, 2), (2, 3), (5, 2)):
        with S.Engine.pipeline_open(p, paths=paths, devices=[0], n_streams=n_streams, chunk_records=1500, sink=L.SINK_TUPLES, n_readers=n_readers) as pl:
            chunks = _collect(pl)
        for i, (cnt, w_h, w_p) in enumerate(want):
            mine = sorted((c for c in chunks if c["src"] == i), key=lambda c: c["first"])
            assert sum(c["n"] for c in mine) == cnt, (n_readers, i)
            assert np.array_equal(np.concatenate([c["hash"] for c in mine]), w_h), (n_readers, i)
            assert np.array_equal(np.concatenate([c["pos"] for c in mine]), w_p), (n_readers, i)


def test_cancel_wakes_a_blocked_consumer(engine):
    """bsk_pipeline_cancel from another thread: the consumer inside bsk_pipeline_next returns (-1), nothing is freed under it, close joins
    the workers afterwards -- what the Go shim's Pipeline.Close does with the goroutine behind Chunks()."""
    import threading
    data, offs = make_reads(20_000, 5, with_n=False)
    p = engine.params(L.MINIMIZER, 21, w=11)
    pl = S.Engine.pipeline_open(p, data=data, offsets=offs, devices=[0], n_streams=2, chunk_records=500, sink=L.SINK_TUPLES, alphabet=L.ALPHA_DNA, repeat=2000)
    got, out = [], {}

    def consumer():
        try:
            for c in pl.chunks():
                got.append(c.sequence)
        except S.PipelineStopped:
            out["stopped"] = True

    t = threading.Thread(target=consumer)
    t.start()
    while len(got) < 3:
        pass
    pl.cancel()
    t.join(30)
    assert not t.is_alive() and out.get("stopped") and got == list(range(len(got)))
    st = pl.close()
    assert st["chunks"] >= len(got)
    # after the natural end cancel is a no-op and close reports the whole run
    pl = S.Engine.pipeline_open(p, data=data, offsets=offs, devices=[0], n_streams=2, chunk_records=5000, sink=L.SINK_COUNTS, alphabet=L.ALPHA_DNA)
    assert len(list(pl.chunks())) == 4
    pl.cancel()
    assert pl.close()["records"] == 20_000


def test_timed_rerun_needs_the_sized_plan(engine):
    """bsk_sketch_timed on an existing result repeats the plan the result was SIZED for -- and refuses another batch or other parameters
    (a fresh plan could write past the arrays: ADVICE round 4)."""
    b1 = engine.synth(L.ALPHA_DNA, 20_000, 150, 7)
    b2 = engine.synth(L.ALPHA_DNA, 20_000, 250, 7)
    p = engine.params(L.MINIMIZER, 21, w=11)
    res, ms = engine.run_timed(b1, p, 1, 2)
    want = res.digest()
    ms2 = (C.c_float * 2)()
    h = C.c_void_p(res.h.value if hasattr(res.h, "value") else res.h)
    assert engine.lib.bsk_sketch_timed(engine.ctx, b1.h, C.byref(p), C.byref(h), 0, 2, ms2) == L.OK
    assert res.digest() == want
    assert engine.lib.bsk_sketch_timed(engine.ctx, b2.h, C.byref(p), C.byref(h), 0, 2, ms2) == L.ERR_ARG
    p2 = engine.params(L.MINIMIZER, 21, w=9)
    assert engine.lib.bsk_sketch_timed(engine.ctx, b1.h, C.byref(p2), C.byref(h), 0, 2, ms2) == L.ERR_ARG
    assert res.digest() == want
