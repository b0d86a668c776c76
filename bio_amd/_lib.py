"""ctypes loader for libbiosketch.so (include/biosketch.h).

The shared object is built in-tree by ``__graft_entry__.build()`` /
``make -C bio_amd/csrc``.  There is no Python or CPU implementation behind it:
if the library is missing or no gfx950 device is visible every compute entry
raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("BSK_LIB") or os.path.join(_HERE, "csrc", "libbiosketch.so")  # BSK_LIB: dev A/B builds

# bsk_err (include/biosketch.h) -- the first eleven are the reference's sentinel errors
OK = 0
ERR_INVALID_K, ERR_EMPTY_SEQ, ERR_SHORT_SEQ, ERR_ILLEGAL_BASE, ERR_K_TOO_LARGE = 1, 2, 3, 4, 5
ERR_INVALID_M, ERR_INVALID_SCALE, ERR_INVALID_S, ERR_INVALID_W, ERR_BUF_NIL, ERR_BUF_NOT_EMPTY = 6, 7, 8, 9, 10, 11
ERR_ARG, ERR_NOMEM, ERR_DEVICE, ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_IO = 64, 65, 66, 67, 68, 69
ERR_NOT_FASTX, ERR_BAD_FASTQ, ERR_STOPPED = 70, 71, 72

KMER, NTHASH, SIMHASH, MINIMIZER, SYNCMER, PROT_HASH, PROT_MINIMIZER = 1, 2, 3, 4, 5, 6, 7
ALPHA_DNA, ALPHA_PROTEIN = 0, 1
# the other nucleotide alphabets of seq/alphabet.go (they only change which letters the two-strand k-mer mode pairs)
ALPHA_DNA_PLAIN, ALPHA_RNA, ALPHA_RNA_REDUNDANT, ALPHA_UNLIMIT = 2, 3, 4, 5

ST_OK, ST_SHORT, ST_ILLEGAL, ST_CODE_MASK = 0x00, 0x01, 0x02, 0x0F
ST_FIRST_WINDOW_TIE, ST_HAS_NON_ACGT = 0x10, 0x20
POS_STRAND_BIT, POS_MASK = 0x80000000, 0x7FFFFFFF


class PipelineStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("records", "bases", "tuples", "chunks", "checksum")] + \
               [(n, C.c_double) for n in ("seconds", "reader_seconds", "reader_wait_seconds", "h2d_pack_seconds", "kernel_seconds", "fetch_seconds")] + \
               [("n_streams", C.c_int32), ("reader_threads", C.c_int32), ("reparsed_pieces", C.c_uint64), ("pin_seconds", C.c_double)]

    def asdict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class PipelineConfig(C.Structure):
    """bsk_pipeline_config (include/biosketch.h)"""
    _fields_ = [("devices", C.POINTER(C.c_int)), ("n_devices", C.c_int32), ("n_streams", C.c_int32), ("chunk_records", C.c_uint64),
                ("sink", C.c_int32), ("sets_scale", C.c_int32), ("alphabet", C.c_int32), ("host_checksum", C.c_int32),
                ("n_readers", C.c_int32), ("reserved", C.c_int32)]


class Chunk(C.Structure):
    """bsk_chunk (include/biosketch.h): one delivered chunk; the arrays are pinned host memory of the pipeline until release"""
    _fields_ = [("sequence", C.c_uint64), ("source_index", C.c_int32), ("device", C.c_int32), ("first_record", C.c_uint64),
                ("n_records", C.c_uint64), ("n_bases", C.c_uint64), ("n_tuples", C.c_uint64), ("n_values", C.c_uint64),
                ("checksum", C.c_uint64), ("link_bytes", C.c_uint64), ("sink", C.c_int32), ("has_pos", C.c_int32),
                ("offsets32", C.c_void_p), ("offsets64", C.c_void_p), ("status", C.c_void_p), ("hash", C.c_void_p),
                ("pos16", C.c_void_p), ("pos32", C.c_void_p), ("opaque", C.c_void_p)]


SINK_COUNTS, SINK_TUPLES, SINK_SETS = 0, 1, 2
CHUNK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(Chunk))


class Params(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("kind", "k", "w", "s", "m", "scale", "canonical", "circular", "codon_table", "frame")]


# every symbol include/biosketch.h declares: (name, restype, argtypes)
_vp = C.c_void_p
_pp = C.POINTER(C.c_void_p)
_u64p = C.POINTER(C.c_uint64)
SYMBOLS = [
    ("bsk_abi_version", C.c_int, []),
    ("bsk_device_count", C.c_int, [C.POINTER(C.c_int)]),
    ("bsk_ctx_create", C.c_int, [C.c_int, _pp]),
    ("bsk_ctx_destroy", None, [_vp]),
    ("bsk_ctx_sync", C.c_int, [_vp]),
    ("bsk_ctx_reload_options", C.c_int, [_vp]),
    ("bsk_build_has_experiments", C.c_int, []),
    ("bsk_last_error", C.c_char_p, [_vp]),
    ("bsk_err_name", C.c_char_p, [C.c_int]),
    ("bsk_batch_from_ascii", C.c_int, [_vp, _vp, _vp, C.c_uint64, C.c_int, _pp]),
    ("bsk_batch_from_packed", C.c_int, [_vp, _vp, C.c_uint64, _vp, C.c_uint64, _pp]),
    ("bsk_batch_synth", C.c_int, [_vp, C.c_int, C.c_uint64, C.c_uint32, C.c_uint64, _pp]),
    ("bsk_batch_info", C.c_int, [_vp, _u64p, _u64p, _u64p, _u64p]),
    ("bsk_batch_fetch_ascii", C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, _vp, C.c_uint64, _vp]),
    ("bsk_batch_translate", C.c_int, [_vp, _vp, C.c_int, C.c_int, _pp]),
    ("bsk_codon_lut", C.c_int, [C.c_int, _vp, C.c_uint64]),
    ("bsk_batch_destroy", None, [_vp]),
    ("bsk_fastx_open", C.c_int, [C.c_char_p, _pp]),
    ("bsk_fastx_read_chunk", C.c_int, [_vp, C.c_uint64, C.c_uint64, _u64p, _pp, _pp, _pp, _pp, _pp]),
    ("bsk_fastx_info", C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("bsk_fastx_error", C.c_char_p, [_vp]),
    ("bsk_fastx_close", None, [_vp]),
    ("bsk_fastx_par_open", C.c_int, [C.c_char_p, C.c_int, C.c_uint64, C.POINTER(_vp)]),
    ("bsk_fastx_par_next", C.c_int, [_vp, C.POINTER(_vp)]),
    ("bsk_fastx_piece_data", C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(_vp), C.POINTER(_vp)]),
    ("bsk_fastx_piece_release", None, [_vp, _vp]),
    ("bsk_fastx_par_info", C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint64)]),
    ("bsk_fastx_par_error", C.c_char_p, [_vp]),
    ("bsk_fastx_par_close", None, [_vp]),
    ("bsk_batch_from_fastx", C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, C.c_int, _pp, _u64p]),
    ("bsk_sketch", C.c_int, [_vp, _vp, C.POINTER(Params), _pp]),
    ("bsk_sketch_timed", C.c_int, [_vp, _vp, C.POINTER(Params), _pp, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    ("bsk_batch_prepare", C.c_int, [_vp, _vp, C.POINTER(Params), C.POINTER(C.c_float)]),
    ("bsk_batch_refill_ascii", C.c_int, [_vp, _pp, _vp, _vp, C.c_uint64, C.c_int]),
    ("bsk_batch_refill_packed", C.c_int, [_vp, _pp, _vp, C.c_uint64, _vp, C.c_uint64]),
    ("bsk_pipeline_fastx", C.c_int, [C.c_int, C.c_char_p, C.c_int, _vp, C.c_int, C.c_uint64, C.c_int, _vp]),
    ("bsk_pipeline_memory", C.c_int, [C.c_int, _vp, _vp, C.c_uint64, C.c_int, _vp, C.c_int, C.c_uint64, C.c_int, C.c_int, _vp]),
    ("bsk_pipeline_fastx_files", C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_uint64, C.c_int, _vp]),
    ("bsk_pipeline_fastx_multi", C.c_int, [_vp, C.c_int, C.c_char_p, C.c_int, _vp, C.c_int, C.c_uint64, C.c_int, _vp]),
    ("bsk_pipeline_memory_multi", C.c_int, [_vp, C.c_int, _vp, _vp, C.c_uint64, C.c_int, _vp, C.c_int, C.c_uint64, C.c_int, C.c_int, _vp]),
    ("bsk_pipeline_trim", None, []),
    ("bsk_pipeline_open_fastx", C.c_int, [C.POINTER(PipelineConfig), C.POINTER(C.c_char_p), C.c_int, C.POINTER(Params), _pp]),
    ("bsk_pipeline_open_memory", C.c_int, [C.POINTER(PipelineConfig), _vp, _vp, C.c_uint64, C.c_int, C.POINTER(Params), _pp]),
    ("bsk_pipeline_next", C.c_int, [_vp, C.POINTER(C.POINTER(Chunk))]),
    ("bsk_pipeline_release", C.c_int, [_vp, C.POINTER(Chunk)]),
    ("bsk_pipeline_close", C.c_int, [_vp, _vp]),
    ("bsk_pipeline_error", C.c_char_p, [_vp]),
    ("bsk_pipeline_cancel", C.c_int, [_vp]),
    ("bsk_pipeline_run", C.c_int, [_vp, CHUNK_FN, _vp, _vp]),
    ("bsk_result_fetch_status", C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, _vp]),
    ("bsk_result_sets_reuse", C.c_int, [_vp, _vp, C.c_int, C.c_int, _pp]),
    ("bsk_sets_fetch_narrow", C.c_int, [_vp, _vp, _vp, _vp, C.c_uint64]),
    ("bsk_comm_unique_id", C.c_int, [_vp]),
    ("bsk_comm_init_rank", C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    ("bsk_comm_init_all", C.c_int, [_pp, C.c_int]),
    ("bsk_gather_counts", C.c_int, [_vp, _vp, C.c_int, _vp]),
    ("bsk_gather_counts_all", C.c_int, [_pp, C.c_int, _vp, C.c_int, _vp]),
    ("bsk_comm_destroy", None, [_vp]),
    ("bsk_result_plan", C.c_int, [_vp, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("bsk_result_class_plan", C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    ("bsk_result_info", C.c_int, [_vp, _u64p, _u64p, C.POINTER(C.c_int)]),
    ("bsk_result_fetch", C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, _vp, _vp, _vp, _vp, C.c_uint64]),
    ("bsk_result_fetch_narrow", C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, _vp, _vp, _vp, _vp, C.c_uint64, _u64p]),
    ("bsk_result_device", C.c_int, [_vp, _pp, _pp, _pp, _pp]),
    ("bsk_result_compact", C.c_int, [_vp, _vp, _pp, _pp, _pp, _u64p]),
    ("bsk_result_device_wide", C.c_int, [_vp, _pp, _pp]),
    ("bsk_result_digest", C.c_int, [_vp, _vp, _u64p, _u64p, _u64p]),
    ("bsk_result_release", None, [_vp]),
    ("bsk_result_sets", C.c_int, [_vp, _vp, C.c_int, C.c_int, _pp]),
    ("bsk_sets_info", C.c_int, [_vp, _u64p, _u64p]),
    ("bsk_sets_fetch", C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, _vp, _vp, C.c_uint64]),
    ("bsk_sets_device", C.c_int, [_vp, _pp, _pp]),
    ("bsk_sets_release", None, [_vp]),
]
SETS_PER_SEQUENCE, SETS_WHOLE_BATCH = 0, 1

_lib = None


class BiosketchUnavailable(RuntimeError):
    pass


def load():
    """Load libbiosketch.so and bind every ABI symbol.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise BiosketchUnavailable(
            f"{SO_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(SO_PATH)
    partial = bool(os.environ.get("BSK_LIB_PARTIAL"))  # dev: a library of ONE translation unit (the sanitized reader, csrc/san-fastx)
    for name, res, args in SYMBOLS:
        if partial and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
