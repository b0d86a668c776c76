"""ctypes binding of the CPU ORACLE (oracle/bio_oracle.c).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by anything under bio_amd/.  The oracle is
a CPU restatement of shenwei356/bio `sketches/` (see bio_oracle.h for pinning).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libbio_oracle.so")

# error codes (bio_oracle.h) -> reference sentinel names (iterator.go:34-53, sketch.go:32-42)
ERR_NAMES = {
    -1: "ErrInvalidK", -2: "ErrEmptySeq", -3: "ErrShortSeq", -4: "ErrIllegalBase",
    -5: "ErrKTooLarge", -6: "ErrInvalidM", -7: "ErrInvalidScale", -8: "ErrInvalidS",
    -9: "ErrInvalidW", -100: "nomem", -101: "capacity",
}
FLAG_FIRST_WINDOW_TIE = 0x10
FLAG_HAS_NON_ACGT = 0x20


class OracleError(Exception):
    def __init__(self, code: int):
        self.code = code
        self.name = ERR_NAMES.get(code, str(code))
        super().__init__(self.name)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "bio_oracle.c")
    hdr = os.path.join(_HERE, "bio_oracle.h")
    if (force or not os.path.exists(_SO)
            or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libbio_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p, u64p, u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
        L.orc_nthash_all.restype = C.c_longlong
        L.orc_nthash_all.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, u64p, u8p, C.c_size_t]
        L.orc_kmer_all.restype = C.c_longlong
        L.orc_kmer_all.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, u64p, C.c_size_t]
        L.orc_kmer_all_alpha.restype = C.c_longlong
        L.orc_kmer_all_alpha.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, u64p, C.c_size_t]
        L.orc_simhash_all.restype = C.c_longlong
        L.orc_simhash_all.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u64p, C.c_size_t]
        for name in ("orc_minimizer_all", "orc_syncmer_all", "orc_minimizer_closed", "orc_syncmer_closed"):
            f = getattr(L, name)
            f.restype = C.c_longlong
            f.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, u64p, u32p, u8p, C.c_size_t,
                          C.POINTER(C.c_uint)]
        L.orc_wyhash.restype = C.c_uint64
        L.orc_wyhash.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64]
        L.orc_protein_hash_all.restype = C.c_longlong
        L.orc_protein_hash_all.argtypes = [C.c_char_p, C.c_size_t, C.c_int, u64p, C.c_size_t]
        for name in ("orc_protein_minimizer_all", "orc_protein_minimizer_closed"):
            f = getattr(L, name)
            f.restype = C.c_longlong
            f.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, u64p, u32p, C.c_size_t, C.POINTER(C.c_uint)]
        L.orc_batch_run.restype = C.c_int
        L.orc_batch_run.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                    u64p, u64p]
        L.orc_seed_fwd.restype = C.c_uint64
        L.orc_seed_fwd.argtypes = [C.c_uint8]
        L.orc_seed_rev.restype = C.c_uint64
        L.orc_seed_rev.argtypes = [C.c_uint8]
        _lib = L
    return _lib


def _b(seq) -> bytes:
    if isinstance(seq, str):
        return seq.encode()
    if isinstance(seq, np.ndarray):
        return seq.tobytes()
    return bytes(seq)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _chk(n):
    if n < 0:
        raise OracleError(int(n))
    return int(n)


def nthash(seq, k, canonical=True, circular=False):
    """NewHashIterator/NextHash (iterator.go:615-665) -> (hashes u64[], strand u8[])."""
    s = _b(seq)
    cap = len(s) + max(k, 1) + 1
    out = np.zeros(cap, np.uint64)
    st = np.zeros(cap, np.uint8)
    n = _chk(lib().orc_nthash_all(s, len(s), k, int(canonical), int(circular), _p(out, C.c_uint64),
                                  _p(st, C.c_uint8), cap))
    return out[:n].copy(), st[:n].copy()


def kmer_codes(seq, k, canonical=True, circular=False, alphabet=0):
    """NewKmerIterator/NextKmer (iterator.go:668-759).  alphabet (numbered as bsk_alphabet: 0 DNAredundant, 2 DNA, 3 RNA,
    4 RNAredundant, 5 Unlimit) = the Seq's Alphabet, whose PairLetter builds the second strand of canonical=False."""
    s = _b(seq)
    cap = 2 * (len(s) + max(k, 1)) + 2
    out = np.zeros(cap, np.uint64)
    n = _chk(lib().orc_kmer_all_alpha(s, len(s), k, int(canonical), int(circular), int(alphabet), _p(out, C.c_uint64), cap))
    return out[:n].copy()


def simhash(seq, k, m, scale, canonical=True, circular=False):
    """NewSimHashIterator/NextSimHash (iterator.go:113-612)."""
    s = _b(seq)
    cap = len(s) + max(k, 1) + 1
    out = np.zeros(cap, np.uint64)
    n = _chk(lib().orc_simhash_all(s, len(s), k, m, scale, int(canonical), int(circular),
                                   _p(out, C.c_uint64), cap))
    return out[:n].copy()


def _sketch(fn, seq, k, x, circular):
    s = _b(seq)
    cap = len(s) + max(k, 1) + 1
    h = np.zeros(cap, np.uint64)
    p = np.zeros(cap, np.uint32)
    st = np.zeros(cap, np.uint8)
    fl = C.c_uint(0)
    n = _chk(fn(s, len(s), k, x, int(circular), _p(h, C.c_uint64), _p(p, C.c_uint32), _p(st, C.c_uint8),
                cap, C.byref(fl)))
    return h[:n].copy(), p[:n].copy(), st[:n].copy(), int(fl.value)


def minimizer(seq, k, w, circular=False, closed=False):
    """NewMinimizerSketch/NextMinimizer (sketch.go:85-138,205-309) -> (hash, pos, strand, flags)."""
    return _sketch(lib().orc_minimizer_closed if closed else lib().orc_minimizer_all, seq, k, w, circular)


def syncmer(seq, k, s, circular=False, closed=False):
    """NewSyncmerSketch/NextSyncmer (sketch.go:142-202,312-477) -> (hash, pos, strand, flags)."""
    return _sketch(lib().orc_syncmer_closed if closed else lib().orc_syncmer_all, seq, k, s, circular)


def wyhash(data, seed=1):
    d = _b(data)
    return int(lib().orc_wyhash(d, len(d), seed))


def protein_hashes(aa, k):
    """NewProteinIterator/Next on protein input (iterator-protein.go:46-90)."""
    s = _b(aa)
    cap = len(s) + 1
    out = np.zeros(cap, np.uint64)
    n = _chk(lib().orc_protein_hash_all(s, len(s), k, _p(out, C.c_uint64), cap))
    return out[:n].copy()


def protein_minimizer(aa, k, w, closed=False):
    """NewProteinMinimizerSketch/Next (sketch-protein.go:62-210) -> (hash, pos, flags)."""
    s = _b(aa)
    cap = len(s) + 1
    h = np.zeros(cap, np.uint64)
    p = np.zeros(cap, np.uint32)
    fl = C.c_uint(0)
    fn = lib().orc_protein_minimizer_closed if closed else lib().orc_protein_minimizer_all
    n = _chk(fn(s, len(s), k, w, _p(h, C.c_uint64), _p(p, C.c_uint32), cap, C.byref(fl)))
    return h[:n].copy(), p[:n].copy(), int(fl.value)


def genetic_code(table):
    """64 amino acids of an NCBI genetic code in TCAG order (seq/codon_tables.go:431-621)."""
    buf = C.create_string_buffer(65)
    if lib().orc_genetic_code(table, buf) != 0:
        raise ValueError(f"unknown codon table {table}")
    return buf.value.decode()


def translate(nt, table=1, frame=1, trim=False, clean=False):
    """CodonTable.Translate(..., allowUnknownCodon=true, markInitCodonAsM=false) (seq/codon_tables.go:205-285)."""
    s = _b(nt)
    cap = len(s) // 3 + 2
    out = np.zeros(cap, np.uint8)
    L = lib()
    L.orc_translate.restype = C.c_longlong
    L.orc_translate.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    n = L.orc_translate(s, len(s), table, frame, int(trim), int(clean), out.ctypes.data, cap)
    if n < 0:
        raise ValueError({-1: "invalid codon table", -2: "invalid frame", -3: "sequence too short to translate"}.get(n, str(n)))
    return out[:n].tobytes().decode("latin-1")


def protein_hashes_nt(nt, k, table=1, frame=1):
    """NewProteinIterator/Next on DNA/RNA input (iterator-protein.go:46-90)."""
    s = _b(nt)
    cap = len(s) // 3 + 2
    out = np.zeros(cap, np.uint64)
    L = lib()
    L.orc_protein_hash_nt.restype = C.c_longlong
    L.orc_protein_hash_nt.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    n = _chk(L.orc_protein_hash_nt(s, len(s), k, table, frame, out.ctypes.data, cap))
    return out[:n].copy()


def protein_minimizer_nt(nt, k, w, table=1, frame=1):
    """NewProteinMinimizerSketch/Next on DNA/RNA input (sketch-protein.go:62-210) -> (hash, pos, flags)."""
    s = _b(nt)
    cap = len(s) // 3 + 2
    h = np.zeros(cap, np.uint64)
    p = np.zeros(cap, np.uint32)
    fl = C.c_uint(0)
    L = lib()
    L.orc_protein_minimizer_nt.restype = C.c_longlong
    L.orc_protein_minimizer_nt.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_size_t, C.c_void_p]
    n = _chk(L.orc_protein_minimizer_nt(s, len(s), k, w, table, frame, h.ctypes.data, p.ctypes.data, cap, C.byref(fl)))
    return h[:n].copy(), p[:n].copy(), int(fl.value)


def batch_run(kind, seqs: np.ndarray, offsets: np.ndarray, k, w_or_s, threads=1):
    """cpu_baseline driver: returns (n_tuples, checksum). kind: 2 nthash, 4 minimizer, 5 syncmer, 7 protmin."""
    seqs = np.ascontiguousarray(seqs, np.uint8)
    offsets = np.ascontiguousarray(offsets, np.uint64)
    nt, ck = C.c_uint64(0), C.c_uint64(0)
    rc = lib().orc_batch_run(kind, seqs.ctypes.data, offsets.ctypes.data, len(offsets) - 1, k, w_or_s,
                             threads, C.byref(nt), C.byref(ck))
    if rc != 0:
        raise OracleError(rc)
    return int(nt.value), int(ck.value)
