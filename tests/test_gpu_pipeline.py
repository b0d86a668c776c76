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
