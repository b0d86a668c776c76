"""CPU: the C-ABI library loads and exports every symbol include/biosketch.h declares; host mirror logic.
No compute calls: this box has no GPU and the engine has no CPU fallback."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "biosketch.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bsk_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_header_symbol():
    from bio_amd import _lib
    if not os.path.exists(_lib.SO_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _lib.load()
    names = _header_functions()
    assert len(names) >= 20
    bound = {n for n, _, _ in _lib.SYMBOLS}
    assert set(names) == bound, (set(names) ^ bound)
    for n in names:
        assert getattr(lib, n) is not None
    assert lib.bsk_abi_version() == 1


def test_no_device_means_loud_failure_not_fallback():
    from bio_amd import _lib, sketches as S
    lib = _lib.load()
    n = C.c_int(-1)
    assert lib.bsk_device_count(C.byref(n)) == 0
    if n.value > 0:
        pytest.skip("a GPU is visible here")
    h = C.c_void_p()
    assert lib.bsk_ctx_create(0, C.byref(h)) == _lib.ERR_NO_DEVICE
    with pytest.raises(S.DeviceError):
        S.Engine(0)


def test_error_names_are_the_reference_sentinels():
    from bio_amd import _lib
    lib = _lib.load()
    exp = {1: "ErrInvalidK", 2: "ErrEmptySeq", 3: "ErrShortSeq", 4: "ErrIllegalBase", 5: "ErrKTooLarge", 6: "ErrInvalidM",
           7: "ErrInvalidScale", 8: "ErrInvalidS", 9: "ErrInvalidW", 10: "ErrBufNil", 11: "ErrBufNotEmpty"}
    for code, name in exp.items():
        assert lib.bsk_err_name(code).decode() == name


def test_cursor_semantics_match_reference_iterators():
    from bio_amd import _lib as L, sketches as S
    pos = np.array([0, 1 | L.POS_STRAND_BIT, 4], np.uint32)
    sk = S.Sketch(0, np.array([11, 22, 33], np.uint64), pos)
    got = []
    while True:
        c, ok = sk.Next()
        if not ok:
            break
        got.append((c, sk.Index(), sk.Strand()))
    assert got == [(11, 0, 0), (22, 1, 1), (33, 4, 0)]
    assert sk.Next() == (0, False)
    it = S.Iterator(0, np.array([5, 6, 7, 8], np.uint64), L.KMER, n_per_strand=2)  # non-canonical: two strands
    idx = []
    while True:
        c, ok, err = it.NextKmer()
        if not ok:
            break
        idx.append(it.Index())
    assert idx == [0, 1, 0, 1] and err is None  # Index() restarts on the reverse strand (iterator.go:720)
    bad = S.Iterator(L.ST_ILLEGAL, np.array([], np.uint64), L.KMER)
    assert bad.NextKmer() == (0, False, S.ErrIllegalBase)


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under bio_amd/ or include/ may import, link or call it."""
    for base in ("bio_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    assert "bio_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, os.path.join(dp, f)
