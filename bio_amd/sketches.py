"""Host-side mirror of the reference ``sketches`` package over the C ABI.

Same names, argument order and error behaviour as shenwei356/bio ``sketches``
(file:line relative to the reference root):

    NewHashIterator(s, k, canonical, circular)      iterator.go:615   -> Iterator.NextHash()   :658
    NewKmerIterator(s, k, canonical, circular)      iterator.go:668   -> Iterator.NextKmer()   :708
    NewSimHashIterator(s, k, m, scale, canon, circ) iterator.go:113   -> Iterator.NextSimHash():191
    NewMinimizerSketch(S, k, w, circular)           sketch.go:85      -> Sketch.NextMinimizer():205
    NewSyncmerSketch(S, k, s, circular)             sketch.go:142     -> Sketch.NextSyncmer()  :312
    NewProteinIterator(s, k, codonTable, frame)     iterator-protein.go:46 -> ProteinIterator.Next() :76
    NewProteinMinimizerSketch(S, k, table, frame, w) sketch-protein.go:62  -> ProteinMinimizerSketch.Next() :106
    Index() on all four types                       iterator.go:776, sketch.go:488, ...

Constructors return ``(obj, err)`` Go-style; ``err`` is one of the sentinel
objects below (compare with ``is``).  The device works on batches: the fast path
is ``Engine.batch(...)`` + ``Engine.run(...)`` whose ``BatchResult.iterator(i)``
hands out the same cursor types; the single-sequence constructors run a batch of
one on the GPU so that code written against the reference keeps working.

Everything computes on the MI355X through libbiosketch.so -- there is no Python
or CPU implementation here.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib as L


# ---- sentinel errors (iterator.go:34-53, sketch.go:32-42) ------------------------------------
class SketchError(Exception):
    def __init__(self, code: int, msg: str):
        super().__init__(msg)
        self.code = code


ErrInvalidK = SketchError(L.ERR_INVALID_K, "sketches: invalid k-mer size")
ErrEmptySeq = SketchError(L.ERR_EMPTY_SEQ, "sketches: empty sequence")
ErrShortSeq = SketchError(L.ERR_SHORT_SEQ, "sketches: sequence too short")
ErrIllegalBase = SketchError(L.ERR_ILLEGAL_BASE, "sketches: illegal base")
ErrKTooLarge = SketchError(L.ERR_K_TOO_LARGE, "sketches: k-mer size is too large")
ErrInvalidM = SketchError(L.ERR_INVALID_M, "sketches: invalid m-mer size, should be in range of [4, k]")
ErrInvalidScale = SketchError(L.ERR_INVALID_SCALE, "sketches: invalid scale, should be in range of [1, k-m+1]")
ErrInvalidS = SketchError(L.ERR_INVALID_S, "kmers: invalid s-mer size")
ErrInvalidW = SketchError(L.ERR_INVALID_W, "kmers: invalid minimimzer window")
ErrBufNil = SketchError(L.ERR_BUF_NIL, "kmers: buffer slice is nil")
ErrBufNotEmpty = SketchError(L.ERR_BUF_NOT_EMPTY, "kmers: buffer has elements")
_SENTINELS = {e.code: e for e in (ErrInvalidK, ErrEmptySeq, ErrShortSeq, ErrIllegalBase, ErrKTooLarge, ErrInvalidM,
                                  ErrInvalidScale, ErrInvalidS, ErrInvalidW, ErrBufNil, ErrBufNotEmpty)}


class DeviceError(RuntimeError):
    """Non-sentinel failure of the engine (device, memory, unsupported)."""


# ---- seq.Seq stand-in (seq/seq.go:29-34): only Alphabet identity and the byte slice cross the boundary
class Alphabet:
    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return self.name


DNA = Alphabet("DNA")
DNAredundant = Alphabet("DNAredundant")
RNA = Alphabet("RNA")
RNAredundant = Alphabet("RNAredundant")
Protein = Alphabet("Protein")
Unlimit = Alphabet("Unlimit")


def _alpha_code(a: Alphabet) -> int:
    """bsk_alphabet of a Seq's Alphabet: the nucleotide alphabets differ only in the letters RevComInplace pairs (iterator.go:719)."""
    return {DNA: L.ALPHA_DNA_PLAIN, RNA: L.ALPHA_RNA, RNAredundant: L.ALPHA_RNA_REDUNDANT, Unlimit: L.ALPHA_UNLIMIT,
            Protein: L.ALPHA_PROTEIN}.get(a, L.ALPHA_DNA)


class Seq:
    def __init__(self, alphabet: Alphabet, seq):
        self.Alphabet = alphabet
        self.Seq = seq.encode() if isinstance(seq, str) else bytes(seq)


def NewSeq(alphabet: Alphabet, seq) -> Tuple[Seq, None]:
    return Seq(alphabet, seq), None


# ---- engine ---------------------------------------------------------------------------------------
class Batch:
    def __init__(self, eng: "Engine", handle):
        self.eng, self.h = eng, handle

    def info(self):
        v = [C.c_uint64() for _ in range(4)]
        self.eng._chk(self.eng.lib.bsk_batch_info(self.h, *[C.byref(x) for x in v]))
        return dict(n_reads=v[0].value, n_bases=v[1].value, device_bytes=v[2].value, n_non_acgt_reads=v[3].value)

    def fetch_ascii(self, first: int, count: int) -> Tuple[np.ndarray, np.ndarray]:
        inf = self.info()
        cap = inf["n_bases"] if count == inf["n_reads"] else None
        offs = np.zeros(count + 1, np.uint64)
        if cap is None:  # two-step: sizes first via a generous bound
            cap = inf["n_bases"]
        buf = np.zeros(max(int(cap), 1), np.uint8)
        self.eng._chk(self.eng.lib.bsk_batch_fetch_ascii(self.eng.ctx, self.h, first, count, buf.ctypes.data, buf.size,
                                                         offs.ctypes.data))
        return buf[: int(offs[-1])].copy(), offs

    def translate(self, codon_table: int = 1, frame: int = 1) -> "Batch":
        """(*seq.Seq).Translate(table, frame, false, false, true, false) of every sequence (seq/seq.go:685) -> protein batch."""
        h = C.c_void_p()
        self.eng._chk(self.eng.lib.bsk_batch_translate(self.eng.ctx, self.h, codon_table, frame, C.byref(h)))
        return Batch(self.eng, h)

    def close(self):
        if self.h:
            self.eng.lib.bsk_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchResult:
    """Device-resident CSR result of one bsk_sketch call."""

    def __init__(self, eng: "Engine", handle, params: L.Params):
        self.eng, self.h, self.params = eng, handle, params
        self._host = None

    def info(self):
        n, t, hp = C.c_uint64(), C.c_uint64(), C.c_int()
        self.eng._chk(self.eng.lib.bsk_result_info(self.h, C.byref(n), C.byref(t), C.byref(hp)))
        return dict(n_reads=n.value, n_tuples=t.value, has_pos=bool(hp.value))

    def plan(self):
        """What ran: the kernel the planner launched for this result, its grid and wavefronts per CU (bsk_result_plan)."""
        name, grid, per_cu = C.c_char_p(), C.c_int(), C.c_int()
        self.eng._chk(self.eng.lib.bsk_result_plan(self.h, C.byref(name), C.byref(grid), C.byref(per_cu)))
        return dict(kernel=(name.value or b"").decode(), grid=grid.value, waves_per_cu=per_cu.value)

    def class_plan(self):
        """bsk_result_class_plan -> (classes besides the bulk, device ms of the passes that cut the batch)"""
        n, ms = C.c_int(), C.c_float()
        self.eng._chk(self.eng.lib.bsk_result_class_plan(self.h, C.byref(n), C.byref(ms)))
        return n.value, float(ms.value)

    def fetch(self, first: int = 0, count: Optional[int] = None):
        """-> (offsets[count+1] rebased, status[count], hash[T], pos[T] or None)"""
        inf = self.info()
        if count is None:
            count = inf["n_reads"] - first
        offs = np.zeros(count + 1, np.uint64)
        status = np.zeros(max(count, 1), np.uint8)
        self.eng._chk(self.eng.lib.bsk_result_fetch(self.eng.ctx, self.h, first, count, offs.ctypes.data,
                                                    status.ctypes.data, None, None, 0))
        T = int(offs[-1])
        hash_ = np.zeros(max(T, 1), np.uint64)
        pos = np.zeros(max(T, 1), np.uint32) if inf["has_pos"] else None
        self.eng._chk(self.eng.lib.bsk_result_fetch(self.eng.ctx, self.h, first, count, offs.ctypes.data,
                                                    status.ctypes.data, hash_.ctypes.data,
                                                    pos.ctypes.data if pos is not None else None, max(T, 1)))
        return offs, status[:count], hash_[:T], (pos[:T] if pos is not None else None)

    def fetch_narrow(self, first: int = 0, count: Optional[int] = None):
        """bsk_result_fetch_narrow -> (offsets[count+1] u32, status[count], hash[T], pos[T] u16 (bit 15 = strand) or None)"""
        inf = self.info()
        if count is None:
            count = inf["n_reads"] - first
        offs = np.zeros(count + 1, np.uint32)
        status = np.zeros(max(count, 1), np.uint8)
        cap = max(int(inf["n_tuples"]), 1)
        hash_ = np.zeros(cap, np.uint64)
        pos = np.zeros(cap, np.uint16) if inf["has_pos"] else None
        nt = C.c_uint64()
        self.eng._chk(self.eng.lib.bsk_result_fetch_narrow(self.eng.ctx, self.h, first, count, offs.ctypes.data, status.ctypes.data, hash_.ctypes.data,
                                                           pos.ctypes.data if pos is not None else None, cap, C.byref(nt)))
        T = int(nt.value)
        return offs, status[:count], hash_[:T], (pos[:T] if pos is not None else None)

    def sets(self, whole_batch: bool = False, scale: int = 1):
        """Sorted distinct hash values per sequence (or of the whole batch), optionally FracMinHash-filtered
        (bsk_result_sets) -> (offsets[n_sets+1], values)."""
        h = C.c_void_p()
        self.eng._opts()
        self.eng._chk(self.eng.lib.bsk_result_sets(self.eng.ctx, self.h, L.SETS_WHOLE_BATCH if whole_batch else L.SETS_PER_SEQUENCE, scale,
                                                   C.byref(h)))
        try:
            ns, nv = C.c_uint64(), C.c_uint64()
            self.eng._chk(self.eng.lib.bsk_sets_info(h, C.byref(ns), C.byref(nv)))
            offs = np.zeros(ns.value + 1, np.uint64)
            vals = np.zeros(max(nv.value, 1), np.uint64)
            self.eng._chk(self.eng.lib.bsk_sets_fetch(self.eng.ctx, h, 0, ns.value, offs.ctypes.data, vals.ctypes.data, vals.size))
            return offs, vals[: nv.value]
        finally:
            self.eng.lib.bsk_sets_release(h)

    def compact(self):
        """bsk_result_compact: dense CSR copy left ON THE DEVICE -> (offsets_ptr, hash_ptr, pos_ptr or None, n_tuples); the arrays belong
        to the engine's context until its next compact()."""
        po, ph, pp = C.c_void_p(), C.c_void_p(), C.c_void_p()
        nt = C.c_uint64()
        self.eng._chk(self.eng.lib.bsk_result_compact(self.eng.ctx, self.h, C.byref(po), C.byref(ph), C.byref(pp), C.byref(nt)))
        return po.value, ph.value, pp.value, int(nt.value)

    def digest(self):
        ck, nt = C.c_uint64(), C.c_uint64()
        sc = (C.c_uint64 * 4)()
        self.eng._chk(self.eng.lib.bsk_result_digest(self.eng.ctx, self.h, C.byref(ck), C.byref(nt), sc))
        return dict(checksum=ck.value, n_tuples=nt.value, short=sc[0], illegal=sc[1], first_window_tie=sc[2],
                    has_non_acgt=sc[3])

    def _cache(self):
        if self._host is None:
            self._host = self.fetch()
        return self._host

    def read(self, i: int):
        """(status, hash[], pos[] or None) of read i -- what the reference iterator over read i yields."""
        offs, status, h, p = self._cache()
        a, b = int(offs[i]), int(offs[i + 1])
        return int(status[i]), h[a:b], (p[a:b] if p is not None else None)

    def close(self):
        if self.h:
            self.eng.lib.bsk_result_release(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """One GPU context (bsk_ctx)."""

    def __init__(self, device: int = 0):
        self.lib = L.load()
        h = C.c_void_p()
        rc = self.lib.bsk_ctx_create(device, C.byref(h))
        if rc != L.OK:
            raise DeviceError(f"bsk_ctx_create({device}) failed: {self.lib.bsk_err_name(rc).decode()}")
        self.ctx = h
        self._opts_seen = self._opts_env()

    @staticmethod
    def _opts_env():
        return tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("BSK_")))

    def _opts(self):
        """The library reads its developer switches (BSK_*) once per context; the test suite flips them inside one process, so this
        mirror reloads them (bsk_ctx_reload_options) when the environment changed since the last call.  Only when BSK_PY_WATCH_ENV
        is set (tests/conftest.py sets it): a production caller pays no scan of its environment per call and uses reload_options()."""
        if not (Engine._watch_env or os.environ.get("BSK_PY_WATCH_ENV")):  # (one dict lookup: set at any time, not only before import)
            return
        now = self._opts_env()
        if now != self._opts_seen:
            self._opts_seen = now
            self.lib.bsk_ctx_reload_options(self.ctx)

    _watch_env = bool(os.environ.get("BSK_PY_WATCH_ENV"))

    def reload_options(self):
        """Re-read the BSK_* developer switches from the environment now (bsk_ctx_reload_options)."""
        self._opts_seen = self._opts_env()
        self.lib.bsk_ctx_reload_options(self.ctx)

    def _chk(self, rc: int):
        if rc == L.OK:
            return
        if rc in _SENTINELS:
            raise _SENTINELS[rc]
        raise DeviceError(f"{self.lib.bsk_err_name(rc).decode()}: {self.lib.bsk_last_error(self.ctx).decode()}")

    # -- multi-GPU: the one collective of the path (bsk_comm_* / bsk_gather_counts: RCCL behind the C ABI)
    def comm_unique_id(self) -> bytes:
        buf = (C.c_uint8 * 128)()
        self._chk(self.lib.bsk_comm_unique_id(buf))
        return bytes(buf)

    def comm_init_rank(self, uid: bytes, rank: int, world: int) -> None:
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        self._chk(self.lib.bsk_comm_init_rank(self.ctx, buf, rank, world))
        self._world = world

    def gather_counts(self, mine: Sequence[int]) -> List[List[int]]:
        """all_gather of this rank's u64 counters; one list per rank, in rank order."""
        world = getattr(self, "_world", 0)
        a = np.asarray(list(mine), np.uint64)
        out = np.zeros(max(world, 1) * len(a), np.uint64)
        self._chk(self.lib.bsk_gather_counts(self.ctx, a.ctypes.data, len(a), out.ctypes.data))
        return [[int(v) for v in row] for row in out.reshape(world, len(a))]

    # -- end to end: file / host memory -> tuples on the host, stages overlapped over n_streams contexts (bsk_pipeline_*)
    @staticmethod
    def pipeline_fastx(path: str, params, n_streams: int = 2, chunk_records: int = 1 << 18, fetch: bool = True, alphabet: int = -1, device: int = 0):
        lib = L.load()
        st = L.PipelineStats()
        rc = lib.bsk_pipeline_fastx(device, path.encode(), alphabet, C.byref(params), n_streams, chunk_records, 1 if fetch else 0, C.byref(st))
        if rc != L.OK:
            raise _SENTINELS.get(rc) or DeviceError(f"bsk_pipeline_fastx: {lib.bsk_err_name(rc).decode()}")
        return st.asdict()

    @staticmethod
    def pipeline_memory_multi(devices, data: np.ndarray, offsets: np.ndarray, params, n_streams: int = 2, chunk_records: int = 1 << 18,
                              repeat: int = 1, fetch: bool = True, alphabet: int = L.ALPHA_DNA):
        """bsk_pipeline_memory over several GPUs of the node (a device may be named more than once): one chunk queue, n_streams workers per device."""
        lib = L.load()
        data = np.ascontiguousarray(data, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        dev = (C.c_int * len(devices))(*devices)
        st = L.PipelineStats()
        rc = lib.bsk_pipeline_memory_multi(dev, len(devices), data.ctypes.data, offsets.ctypes.data, len(offsets) - 1, alphabet, C.byref(params),
                                           n_streams, chunk_records, repeat, 1 if fetch else 0, C.byref(st))
        if rc != L.OK:
            raise _SENTINELS.get(rc) or DeviceError(f"bsk_pipeline_memory_multi: {lib.bsk_err_name(rc).decode()}")
        return st.asdict()

    @staticmethod
    def pipeline_fastx_multi(devices, path: str, params, n_streams: int = 2, chunk_records: int = 1 << 18, fetch: bool = True, alphabet: int = -1):
        lib = L.load()
        dev = (C.c_int * len(devices))(*devices)
        st = L.PipelineStats()
        rc = lib.bsk_pipeline_fastx_multi(dev, len(devices), path.encode(), alphabet, C.byref(params), n_streams, chunk_records, 1 if fetch else 0, C.byref(st))
        if rc != L.OK:
            raise _SENTINELS.get(rc) or DeviceError(f"bsk_pipeline_fastx_multi: {lib.bsk_err_name(rc).decode()}")
        return st.asdict()

    @staticmethod
    def pipeline_open(params, *, paths=None, data: np.ndarray = None, offsets: np.ndarray = None, devices=(0,), n_streams: int = 2,
                      chunk_records: int = 1 << 18, sink: int = L.SINK_TUPLES, sets_scale: int = 1, alphabet: int = -1, repeat: int = 1,
                      host_checksum: bool = False, n_readers: int = 0) -> "Pipeline":
        """bsk_pipeline_open_fastx / _memory: the pipeline with a consumer (the role of fastx's ChunkChan, seqio/fastx/reader.go:562-608)."""
        return Pipeline(params, paths, data, offsets, devices, n_streams, chunk_records, sink, sets_scale, alphabet, repeat, host_checksum, n_readers)

    @staticmethod
    def pipeline_trim() -> None:
        """Return the pinned host buffers the pipelines pool between calls (bsk_pipeline_trim)."""
        L.load().bsk_pipeline_trim()

    @staticmethod
    def pipeline_fastx_files(paths, params, n_streams: int = 3, n_readers: int = 0, chunk_records: int = 1 << 18, fetch: bool = True,
                             alphabet: int = -1, device: int = 0):
        """Several FASTA/FASTQ files (plain or gzip) through one pipeline, n_readers of them read at once."""
        lib = L.load()
        st = L.PipelineStats()
        arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
        rc = lib.bsk_pipeline_fastx_files(device, arr, len(paths), alphabet, C.byref(params), n_streams, n_readers, chunk_records, 1 if fetch else 0,
                                          C.byref(st))
        if rc != L.OK:
            raise _SENTINELS.get(rc) or DeviceError(f"bsk_pipeline_fastx_files: {lib.bsk_err_name(rc).decode()}")
        return st.asdict()

    @staticmethod
    def pipeline_memory(data: np.ndarray, offsets: np.ndarray, params, n_streams: int = 2, chunk_records: int = 1 << 20, repeat: int = 1,
                        fetch: bool = True, alphabet: int = L.ALPHA_DNA, device: int = 0):
        lib = L.load()
        data = np.ascontiguousarray(data, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        st = L.PipelineStats()
        rc = lib.bsk_pipeline_memory(device, data.ctypes.data, offsets.ctypes.data, len(offsets) - 1, alphabet, C.byref(params), n_streams,
                                     chunk_records, repeat, 1 if fetch else 0, C.byref(st))
        if rc != L.OK:
            raise _SENTINELS.get(rc) or DeviceError(f"bsk_pipeline_memory: {lib.bsk_err_name(rc).decode()}")
        return st.asdict()

    # -- batches
    def batch(self, seqs: Sequence, alphabet: int = L.ALPHA_DNA) -> Batch:
        bs = [s.Seq if isinstance(s, Seq) else (s.encode() if isinstance(s, str) else bytes(s)) for s in seqs]
        offs = np.zeros(len(bs) + 1, np.uint64)
        if bs:
            offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
        data = np.frombuffer(b"".join(bs), np.uint8) if offs[-1] else np.zeros(1, np.uint8)
        return self.batch_from_arrays(data, offs, alphabet)

    def batch_from_arrays(self, data: np.ndarray, offsets: np.ndarray, alphabet: int = L.ALPHA_DNA) -> Batch:
        data = np.ascontiguousarray(data, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        h = C.c_void_p()
        self._opts()
        self._chk(self.lib.bsk_batch_from_ascii(self.ctx, data.ctypes.data, offsets.ctypes.data, len(offsets) - 1,
                                                alphabet, C.byref(h)))
        return Batch(self, h)

    def batch_from_packed(self, words: np.ndarray, desc: np.ndarray) -> Batch:
        words = np.ascontiguousarray(words, np.uint32)
        desc = np.ascontiguousarray(desc, np.uint64)
        h = C.c_void_p()
        self._opts()
        self._chk(self.lib.bsk_batch_from_packed(self.ctx, words.ctypes.data, len(words), desc.ctypes.data, len(desc),
                                                 C.byref(h)))
        return Batch(self, h)

    def batch_from_fastx(self, reader, max_records: int = 0, max_bytes: int = 0, alphabet: int = -1):
        """Next chunk of a bio_amd.fastx.Reader straight into a device batch (bsk_batch_from_fastx): (Batch or None, n_records).
        alphabet < 0: the reader's guess from its first record, as the reference does."""
        h = C.c_void_p()
        n = C.c_uint64()
        self._opts()
        self._chk(self.lib.bsk_batch_from_fastx(self.ctx, reader.h, max_records, max_bytes, alphabet, C.byref(h), C.byref(n)))
        return (Batch(self, h) if n.value else None), n.value

    def synth(self, alphabet: int, n: int, length: int, seed: int) -> Batch:
        h = C.c_void_p()
        self._opts()
        self._chk(self.lib.bsk_batch_synth(self.ctx, alphabet, n, length, seed, C.byref(h)))
        return Batch(self, h)

    # -- compute
    @staticmethod
    def params(kind, k, w=0, s=0, m=0, scale=0, canonical=True, circular=False, codon_table=1, frame=1) -> L.Params:
        return L.Params(kind, k, w, s, m, scale, int(bool(canonical)), int(bool(circular)), codon_table, frame)

    def run(self, batch: Batch, p: L.Params, reuse: Optional[BatchResult] = None) -> BatchResult:
        h = reuse.h if reuse is not None else C.c_void_p()
        self._opts()
        self._chk(self.lib.bsk_sketch(self.ctx, batch.h, C.byref(p), C.byref(h)))
        if reuse is not None:
            reuse.h, reuse.params, reuse._host = h, p, None
            return reuse
        return BatchResult(self, h, p)

    def run_timed(self, batch: Batch, p: L.Params, warmup: int, iters: int, reuse: Optional[BatchResult] = None):
        h = reuse.h if reuse is not None else C.c_void_p()
        ms = (C.c_float * max(iters, 1))()
        self._opts()
        self._chk(self.lib.bsk_sketch_timed(self.ctx, batch.h, C.byref(p), C.byref(h), warmup, iters, ms))
        res = reuse if reuse is not None else BatchResult(self, h, p)
        res.h, res.params, res._host = h, p, None
        return res, [ms[i] for i in range(iters)]

    def prepare(self, batch: Batch, p: L.Params) -> float:
        """Per-batch preparation a sketch with these parameters would do on its first call (length-binned units of a ragged batch),
        done now; returns the device milliseconds of the pass (0.0: the plan needs none).  bsk_batch_prepare."""
        ms = C.c_float(0.0)
        self._opts()
        self._chk(self.lib.bsk_batch_prepare(self.ctx, batch.h, C.byref(p), C.byref(ms)))
        return float(ms.value)

    def sync(self):
        self._chk(self.lib.bsk_ctx_sync(self.ctx))

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.bsk_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_engine: Optional[Engine] = None


def default_engine() -> Engine:
    global _default_engine
    if _default_engine is None:
        _default_engine = Engine(0)
    return _default_engine


# ---- cursors: the reference's iterator types over one read's slice of a BatchResult -----------------
class _Cursor:
    def __init__(self, status: int, codes: np.ndarray, pos: Optional[np.ndarray]):
        self._status, self._codes, self._pos = status, codes, pos
        self._i = 0
        self._idx = -1

    def _next(self):
        if self._i >= len(self._codes):
            return 0, False
        c = int(self._codes[self._i])
        self._idx = int(self._pos[self._i] & L.POS_MASK) if self._pos is not None else self._i
        self._strand = int(self._pos[self._i] >> 31) if self._pos is not None else 0
        self._i += 1
        return c, True

    def Index(self) -> int:
        return self._idx

    def flags(self) -> int:
        """engine extension: the read's BSK_ST_* status byte"""
        return self._status


class Iterator(_Cursor):
    """k-mer code / ntHash / SimHash iterator (iterator.go:60-106)."""

    def __init__(self, status, codes, kind, n_per_strand=None):
        super().__init__(status, codes, None)
        self._kind = kind
        self._nps = n_per_strand

    def NextHash(self):  # iterator.go:658
        return self._next()

    def NextSimHash(self):  # iterator.go:191
        return self._next()

    def NextKmer(self):  # iterator.go:708 -> (code, ok, err)
        c, ok = self._next()
        if not ok and (self._status & L.ST_CODE_MASK) == L.ST_ILLEGAL:
            return 0, False, ErrIllegalBase
        if ok and self._nps:  # non-canonical: Index() restarts at 0 on the reverse strand (iterator.go:720)
            self._idx = self._idx % self._nps
        return c, ok, None

    def Next(self):  # iterator.go:762
        if self._kind == L.KMER:
            return self.NextKmer()
        c, ok = self._next()
        return c, ok, None


class Sketch(_Cursor):
    """minimizer / syncmer sketch iterator (sketch.go:45-77)."""

    def NextMinimizer(self):  # sketch.go:205
        return self._next()

    def NextSyncmer(self):  # sketch.go:312
        return self._next()

    def Next(self):  # sketch.go:480
        return self._next()

    def Strand(self) -> int:
        """engine extension: 1 iff the reverse-strand hash was the canonical one for the last tuple"""
        return self._strand


class ProteinIterator(_Cursor):
    def Next(self):  # iterator-protein.go:76
        return self._next()


class ProteinMinimizerSketch(_Cursor):
    def Next(self):  # sketch-protein.go:106
        return self._next()


class ChunkView:
    """One chunk of a Pipeline: numpy VIEWS over the pipeline's pinned arrays (no copy is made on delivery), valid until the next chunk is
    taken -- copy what you keep.  offsets / status / hash are views; pos / strand are decoded from the narrow arrays on first use."""

    def __init__(self, c: L.Chunk):
        self.sequence, self.source_index, self.device = c.sequence, c.source_index, c.device
        self.first_record, self.n_records, self.n_bases, self.n_tuples, self.n_values = c.first_record, c.n_records, c.n_bases, c.n_tuples, c.n_values
        self.checksum, self.link_bytes, self.sink, self.has_pos = c.checksum, c.link_bytes, c.sink, bool(c.has_pos)
        n, nv = int(c.n_records), int(c.n_values)

        def view(ptr, count, dtype):
            if not ptr:
                return None
            if count == 0:
                return np.empty(0, dtype)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape=(count,))

        self.status = view(c.status, n, np.uint8)
        self.offsets = view(c.offsets32, n + 1, np.uint32) if c.offsets32 else view(c.offsets64, n + 1, np.uint64)
        self.hash = view(c.hash, nv, np.uint64)
        self.pos16 = view(c.pos16, nv, np.uint16) if nv else None   # BSK_POS16 encoding (bit 15 = strand), as delivered
        self.pos32 = view(c.pos32, nv, np.uint32) if nv else None   # BSK_POS encoding (bit 31 = strand): chunks with a read of 32 768 bases or more
        self._pos = self._strand = None

    @property
    def pos(self):
        """positions (u32, strand bit removed), decoded on first use; None for kinds with implicit positions / sets"""
        if self._pos is None:
            if self.pos16 is not None:
                self._pos = (self.pos16 & 0x7FFF).astype(np.uint32)
            elif self.pos32 is not None:
                self._pos = self.pos32 & L.POS_MASK
        return self._pos

    @property
    def strand(self):
        if self._strand is None:
            if self.pos16 is not None:
                self._strand = (self.pos16 >> 15).astype(np.uint8)
            elif self.pos32 is not None:
                self._strand = (self.pos32 >> 31).astype(np.uint8)
        return self._strand


class PipelineStopped(Exception):
    """bsk_pipeline_next after bsk_pipeline_cancel / an early close: the run was stopped, not failed."""


class Pipeline:
    """bsk_pipeline: sketches of every chunk of the input, delivered in input order (include/biosketch.h, "the pipeline with a consumer").

        with Engine.pipeline_open(params, paths=["reads.fq"], sink=L.SINK_SETS, sets_scale=100) as pl:
            for chunk in pl.chunks():
                ...chunk.offsets / chunk.hash / chunk.pos / chunk.status...
        pl.stats
    """

    def __init__(self, params, paths, data, offsets, devices, n_streams, chunk_records, sink, sets_scale, alphabet, repeat, host_checksum, n_readers):
        self.lib = L.load()
        self._dev = (C.c_int * len(devices))(*devices)
        cfg = L.PipelineConfig(self._dev, len(devices), n_streams, chunk_records, sink, sets_scale, alphabet, 1 if host_checksum else 0, n_readers, 0)
        self.h = C.c_void_p()
        self._keep = (data, offsets, params)
        if paths is not None:
            arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
            rc = self.lib.bsk_pipeline_open_fastx(C.byref(cfg), arr, len(paths), C.byref(params), C.byref(self.h))
        else:
            data = np.ascontiguousarray(data, dtype=np.uint8)
            offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
            self._keep = (data, offsets, params)
            rc = self.lib.bsk_pipeline_open_memory(C.byref(cfg), data.ctypes.data, offsets.ctypes.data, len(offsets) - 1, repeat, C.byref(params), C.byref(self.h))
        if rc != L.OK:
            self.h = C.c_void_p()
            raise _SENTINELS.get(rc) or DeviceError(f"bsk_pipeline_open: {self.lib.bsk_err_name(rc).decode()}")
        self.stats = None
        self._held = None

    def next(self) -> Optional[ChunkView]:
        """The next chunk in input order, None at the end.  Releases the chunk handed out before."""
        self._release()
        c = C.POINTER(L.Chunk)()
        rc = self.lib.bsk_pipeline_next(self.h, C.byref(c))
        if rc == -1:
            raise PipelineStopped("the pipeline was cancelled")
        if rc != L.OK:
            msg = self.lib.bsk_pipeline_error(self.h).decode()
            raise _SENTINELS.get(rc) or DeviceError(f"bsk_pipeline_next: {self.lib.bsk_err_name(rc).decode()}: {msg}")
        if not c:
            return None
        self._held = c
        return ChunkView(c.contents)

    def _release(self):
        if self._held is not None:
            self.lib.bsk_pipeline_release(self.h, self._held)
            self._held = None

    def chunks(self):
        while True:
            c = self.next()
            if c is None:
                return
            yield c

    def cancel(self) -> None:
        """Stops the run from any thread and frees nothing (bsk_pipeline_cancel): a consumer blocked in next() gets PipelineStopped."""
        if self.h:
            self.lib.bsk_pipeline_cancel(self.h)

    def close(self):
        if self.h:
            self._release()
            st = L.PipelineStats()
            rc = self.lib.bsk_pipeline_close(self.h, C.byref(st))
            self.h = C.c_void_p()
            self.stats = st.asdict()
            if rc not in (L.OK, -1):
                raise _SENTINELS.get(rc) or DeviceError(f"bsk_pipeline_close: {self.lib.bsk_err_name(rc).decode()}")
        return self.stats

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def _single(seq: Seq, p: L.Params, alphabet: int, eng: Optional[Engine]):
    eng = eng or default_engine()
    b = eng.batch([seq], alphabet)
    try:
        res = eng.run(b, p)
    except SketchError as e:
        b.close()
        return None, None, e
    status, codes, pos = res.read(0)
    res.close()
    b.close()
    if (status & L.ST_CODE_MASK) == L.ST_SHORT:
        return None, None, ErrShortSeq
    return (status, codes, pos), None, None


def NewHashIterator(s: Seq, k: int, canonical: bool, circular: bool, engine: Optional[Engine] = None):
    got, _, err = _single(s, Engine.params(L.NTHASH, k, canonical=canonical, circular=circular), L.ALPHA_DNA, engine)
    if err is not None:
        return None, err
    return Iterator(got[0], got[1], L.NTHASH), None


def NewKmerIterator(s: Seq, k: int, canonical: bool, circular: bool, engine: Optional[Engine] = None):
    alpha = _alpha_code(s.Alphabet)
    got, _, err = _single(s, Engine.params(L.KMER, k, canonical=canonical, circular=circular), L.ALPHA_DNA if alpha == L.ALPHA_PROTEIN else alpha, engine)
    if err is not None:
        return None, err
    nps = None if canonical else (len(s.Seq) + (k - 1 if circular else 0) - k + 1)
    return Iterator(got[0], got[1], L.KMER, nps), None


def NewSimHashIterator(s: Seq, k: int, m: int, scale: int, canonical: bool, circular: bool,
                       engine: Optional[Engine] = None):
    got, _, err = _single(s, Engine.params(L.SIMHASH, k, m=m, scale=scale, canonical=canonical, circular=circular),
                          L.ALPHA_DNA, engine)
    if err is not None:
        return None, err
    return Iterator(got[0], got[1], L.SIMHASH), None


def NewMinimizerSketch(S: Seq, k: int, w: int, circular: bool, engine: Optional[Engine] = None):
    got, _, err = _single(S, Engine.params(L.MINIMIZER, k, w=w, circular=circular), L.ALPHA_DNA, engine)
    if err is not None:
        return None, err
    return Sketch(*got), None


def NewSyncmerSketch(S: Seq, k: int, s: int, circular: bool, engine: Optional[Engine] = None):
    got, _, err = _single(S, Engine.params(L.SYNCMER, k, s=s, circular=circular), L.ALPHA_DNA, engine)
    if err is not None:
        return None, err
    return Sketch(*got), None


def NewProteinIterator(s: Seq, k: int, codonTable: int, frame: int, engine: Optional[Engine] = None):
    alpha = L.ALPHA_PROTEIN if s.Alphabet is Protein else L.ALPHA_DNA
    got, _, err = _single(s, Engine.params(L.PROT_HASH, k, codon_table=codonTable, frame=frame), alpha, engine)
    if err is not None:
        return None, err
    return ProteinIterator(got[0], got[1], None), None


def NewProteinMinimizerSketch(S: Seq, k: int, codonTable: int, frame: int, w: int, engine: Optional[Engine] = None):
    alpha = L.ALPHA_PROTEIN if S.Alphabet is Protein else L.ALPHA_DNA
    if k >= 1 and len(S.Seq) < k * 3:  # upstream's order: k, then this length check (sketch-protein.go:66), only then w (:69)
        return None, ErrShortSeq
    got, _, err = _single(S, Engine.params(L.PROT_MINIMIZER, k, w=w, codon_table=codonTable, frame=frame), alpha, engine)
    if err is not None:
        return None, err
    return ProteinMinimizerSketch(*got), None
