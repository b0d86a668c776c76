"""Python mirror of seqio/fastx (Reader.Read / ChunkChan, reader.go:233-369,562-608) over the C ABI's host-side reader.

    reader, err = NewDefaultReader("reads.fq.gz")
    while True:
        record, err = reader.Read()          # (None, EOF) at the end, like io.EOF upstream
        if err is not None: break
        record.Name, record.ID, record.Seq.Seq, record.Seq.Qual

    for chunk in reader.chunks(100000):      # the batching form the GPU path wants (ChunkChan)
        batch = engine.batch_from_arrays(chunk.seq, chunk.offsets, chunk.alphabet)

Names follow the reference (Record.ID = first word of the header, DefaultIDRegexp `^(\\S+)\\s?`, reader.go:109).
"""
from __future__ import annotations

import ctypes as C
import re
from typing import Iterator, Optional

import numpy as np

from . import _lib as L
from .sketches import DNA, DNAredundant, RNA, RNAredundant, Protein, Unlimit, Seq


class FastxError(Exception):
    def __init__(self, code: int, msg: str):
        super().__init__(msg)
        self.code = code


EOF = FastxError(0, "EOF")
ErrNotFASTXFormat = FastxError(70, "fastx: invalid FASTA/Q format")   # reader.go:16
ErrBadFASTQFormat = FastxError(71, "fastx: bad fastq format")         # reader.go:19 (also ErrUnequalSeqAndQual :22)
_ERRS = {70: ErrNotFASTXFormat, 71: ErrBadFASTQFormat}
_ID = re.compile(rb"^(\S+)\s?")


class Record:
    """fastx.Record (records.go:13-18): ID, Name, Desc, Seq (with Qual for FASTQ)."""

    def __init__(self, name: bytes, seq: bytes, qual: Optional[bytes], alphabet):
        self.Name = name
        m = _ID.match(name)
        self.ID = m.group(1) if m else name
        self.Desc = name[m.end():] if m else b""
        self.Seq = Seq(alphabet, seq)
        self.Seq.Qual = qual if qual is not None else b""


class Chunk:
    def __init__(self, seq, offsets, names, name_offsets, qual, alphabet):
        self.seq, self.offsets, self.names, self.name_offsets, self.qual, self.alphabet = seq, offsets, names, name_offsets, qual, alphabet

    def __len__(self):
        return len(self.offsets) - 1

    def name(self, i: int) -> bytes:
        return self.names[int(self.name_offsets[i]):int(self.name_offsets[i + 1])].tobytes()

    def sequence(self, i: int) -> bytes:
        return self.seq[int(self.offsets[i]):int(self.offsets[i + 1])].tobytes()

    def quality(self, i: int) -> Optional[bytes]:
        return None if self.qual is None else self.qual[int(self.offsets[i]):int(self.offsets[i + 1])].tobytes()


class ParallelReader:
    """Block-parallel reader for plain files (bsk_fastx_par_*): the serial reader's records in its order, sequences only."""

    def __init__(self, path: str, threads: int = 4, piece_bytes: int = 0):
        self.lib = L.load()
        self.h = C.c_void_p()
        rc = self.lib.bsk_fastx_par_open(path.encode(), threads, piece_bytes, C.byref(self.h))
        if rc != 0:
            self.h = None
            raise _ERRS.get(rc, FastxError(rc, f"cannot open {path} for block-parallel reading"))

    def pieces(self) -> Iterator[Tuple[np.ndarray, np.ndarray]]:
        """(sequence bytes, offsets[n+1]) of every piece; raises FastxError where the serial reader would."""
        while True:
            pc = C.c_void_p()
            rc = self.lib.bsk_fastx_par_next(self.h, C.byref(pc))
            if rc != 0:
                raise _ERRS.get(rc, FastxError(rc, self.lib.bsk_fastx_par_error(self.h).decode()))
            if not pc.value:
                return
            n, sb, so = C.c_uint64(), C.c_void_p(), C.c_void_p()
            self.lib.bsk_fastx_piece_data(pc, C.byref(n), C.byref(sb), C.byref(so))
            offs = np.ctypeslib.as_array(C.cast(so, C.POINTER(C.c_uint64)), (n.value + 1,)).copy()
            seq = np.ctypeslib.as_array(C.cast(sb, C.POINTER(C.c_uint8)), (max(int(offs[-1]), 1),))[: int(offs[-1])].copy()
            self.lib.bsk_fastx_piece_release(self.h, pc)
            yield seq, offs

    def sequences(self) -> list:
        out = []
        for seq, offs in self.pieces():
            out += [seq[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(len(offs) - 1)]
        return out

    def info(self):
        q, a, r = C.c_int(), C.c_int(), C.c_uint64()
        self.lib.bsk_fastx_par_info(self.h, C.byref(q), C.byref(a), C.byref(r))
        return dict(is_fastq=q.value, alphabet=a.value, reparsed_pieces=r.value)

    def close(self):
        if self.h:
            self.lib.bsk_fastx_par_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Reader:
    def __init__(self, path: str):
        self.lib = L.load()
        self.h = C.c_void_p()
        rc = self.lib.bsk_fastx_open(path.encode(), C.byref(self.h))
        if rc != 0:
            raise OSError(f"fastx: cannot open {path}")
        self._pending: list = []
        self._err: Optional[FastxError] = None

    # -- chunked form
    def read_chunk(self, max_records: int = 0, max_bytes: int = 0) -> Optional[Chunk]:
        """Next chunk, None at end of file; raises FastxError on a format error."""
        n = C.c_uint64()
        sb, so, nb, no, qb = (C.c_void_p() for _ in range(5))
        rc = self.lib.bsk_fastx_read_chunk(self.h, max_records, max_bytes, C.byref(n), C.byref(sb), C.byref(so), C.byref(nb), C.byref(no),
                                           C.byref(qb))
        if rc != 0:
            raise _ERRS.get(rc, FastxError(rc, self.lib.bsk_fastx_error(self.h).decode()))
        if n.value == 0:
            return None
        cnt = n.value
        offs = np.ctypeslib.as_array(C.cast(so, C.POINTER(C.c_uint64)), (cnt + 1,)).copy()
        noffs = np.ctypeslib.as_array(C.cast(no, C.POINTER(C.c_uint64)), (cnt + 1,)).copy()
        seq = np.ctypeslib.as_array(C.cast(sb, C.POINTER(C.c_uint8)), (max(int(offs[-1]), 1),))[: int(offs[-1])].copy()
        names = np.ctypeslib.as_array(C.cast(nb, C.POINTER(C.c_uint8)), (max(int(noffs[-1]), 1),))[: int(noffs[-1])].copy()
        qual = None
        if qb.value:
            qual = np.ctypeslib.as_array(C.cast(qb, C.POINTER(C.c_uint8)), (max(int(offs[-1]), 1),))[: int(offs[-1])].copy()
        return Chunk(seq, offs, names, noffs, qual, self.alphabet)

    def chunks(self, max_records: int = 0, max_bytes: int = 0) -> Iterator[Chunk]:
        while True:
            c = self.read_chunk(max_records, max_bytes)
            if c is None:
                return
            yield c

    @property
    def IsFastq(self) -> bool:
        q, a = C.c_int(), C.c_int()
        self.lib.bsk_fastx_info(self.h, C.byref(q), C.byref(a))
        return q.value == 1

    @property
    def alphabet(self) -> int:
        q, a = C.c_int(), C.c_int()
        self.lib.bsk_fastx_info(self.h, C.byref(q), C.byref(a))
        return a.value

    # -- record form (Reader.Read, reader.go:233)
    def Read(self):
        if not self._pending and self._err is None:
            try:
                c = self.read_chunk(4096)
            except FastxError as e:
                self._err = e
                c = None
            if c is None and self._err is None:
                self._err = EOF
            if c is not None:
                # the reader hands out the guessed alphabet itself (reader.go:430-435): the two-strand k-mer mode pairs
                # letters with the sequence's own alphabet, so DNAredundant / RNA / RNAredundant / Unlimit must not become DNA
                ab = {L.ALPHA_DNA_PLAIN: DNA, L.ALPHA_DNA: DNAredundant, L.ALPHA_RNA: RNA, L.ALPHA_RNA_REDUNDANT: RNAredundant,
                      L.ALPHA_PROTEIN: Protein}.get(c.alphabet, Unlimit)
                self._pending = [Record(c.name(i), c.sequence(i), c.quality(i), ab) for i in range(len(c))][::-1]
        if self._pending:
            return self._pending.pop(), None
        return None, self._err

    def Close(self):
        if self.h:
            self.lib.bsk_fastx_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.Close()
        except Exception:
            pass


def NewDefaultReader(path: str):
    """fastx.NewDefaultReader (reader.go:112): (reader, err)."""
    try:
        return Reader(path), None
    except OSError as e:
        return None, e
