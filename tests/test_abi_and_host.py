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


def test_codon_tables_of_the_library_equal_the_reference_matrix(oracle):
    """bsk_codon_lut (host-only; built as "every expansion agrees") == the 16x16x16 matrix of codonTableFromText
    (three passes, seq/codon_tables.go:317-429) with empty entries read as 'X' (Get, :172-174), for all 24 codes."""
    from bio_amd import _lib as L
    lib = L.load()
    import json
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "codon_golden.json")))
    olib = oracle.lib()
    for tid in sorted(int(t) for t in gold["ncbieaa"]):
        lut = np.zeros(4416, np.uint8)
        assert lib.bsk_codon_lut(tid, lut.ctypes.data, lut.size) == 0
        m = np.zeros(4096, np.uint8)
        assert olib.orc_codon_matrix(tid, m.ctypes.data_as(C.c_void_p)) == 0
        want = np.where(m == 0, ord("X"), m).astype(np.uint8)
        assert np.array_equal(lut[:4096], want), tid
        aa = gold["ncbieaa"][str(tid)]
        tcag = "TCAG"
        for c in range(64):  # 2-bit table: A0 C1 G2 T3, first base most significant
            codon = "".join("ACGT"[(c >> s) & 3] for s in (4, 2, 0))
            idx = tcag.index(codon[0]) * 16 + tcag.index(codon[1]) * 4 + tcag.index(codon[2])
            assert chr(lut[4352 + c]) == aa[idx], (tid, codon)
        for b in range(256):  # letter -> IUPAC set
            want_set = {"A": 1, "C": 2, "G": 4, "T": 8, "U": 8, "N": 15, "M": 3, "R": 5, "W": 9, "S": 6, "Y": 10, "K": 12, "V": 7,
                        "H": 11, "D": 13, "B": 14, " ": 0, "*": 0, "-": 0}.get(chr(b).upper(), 16)
            assert lut[4096 + b] == want_set, (tid, b)
    assert lib.bsk_codon_lut(7, np.zeros(4416, np.uint8).ctypes.data, 4416) != 0      # no such genetic code
    assert lib.bsk_codon_lut(1, np.zeros(16, np.uint8).ctypes.data, 16) != 0          # buffer too small


def test_hand_written_asm_is_safe():
    """scripts/check_asm.py: every inline-asm block with an SCC-writing SALU op names the clobber (the round-2 bug class), and in the
    ISA of EVERY shipped instantiation of k_minimizer_pk / k_minimizer_ring (w = 2..13) and k_syncmer_pk / k_syncmer_pkl (k - s = 4..20 / 4..24) nothing touches
    the registers of a hidden (inline-asm) load before the hand-written wait, nor a register the ring kernel reserves (w = 15, 16 of the
    packed kernel once spilled in-flight registers: the check is width-dependent).  45 hipcc -S runs in parallel, ~40 s on 8 cores."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "scripts", "check_asm.py"), "all"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr


def test_developer_switches_are_read_once_per_context():
    """No getenv on the bsk_sketch path: the host translation units read the BSK_* switches only in BskOpts::load (bsk_ctx_create,
    bsk_ctx_reload_options; biosketch.hip)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for fn in ("biosketch.hip", "planner.hip", "launch.hip", "tiles.hip", "classes.hip", "sets.hip", "comm.cpp"):
        text = open(os.path.join(root, "bio_amd", "csrc", fn)).read()
        body = text
        if fn == "biosketch.hip":  # the two functions that may call getenv
            a, b = text.index("u32 env_u32("), text.index('extern "C" int bsk_ctx_reload_options')
            body = text[:a] + text[b:]
        assert "getenv" not in body, fn


def test_planner_table_is_generated_from_the_committed_sweep():
    """bio_amd/csrc/planner_table.hpp (every threshold of the planner, with the measurement it came from) is what scripts/fit_planner.py
    makes of profiles/r06/planner_sweep.jsonl -- an edited header or a changed sweep file without the regeneration fails here -- and the
    planner's sources carry no ring / slab / syncmer / tile threshold of their own any more."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "scripts", "fit_planner.py"), "--check"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    src = "".join(open(os.path.join(root, "bio_amd", "csrc", fn)).read() for fn in ("planner.hip", "launch.hip", "tiles.hip", "classes.hip", "biosketch.hip"))
    assert "PlannerTable::ring_cap" in src and "17.0 + 2.5 * p->w" not in src and "* 2.6 / (p->w + 1.0)" not in src
    hdr = open(os.path.join(root, "bio_amd", "csrc", "planner_table.hpp")).read()
    for name in ("ring_cap", "dense_min", "slab_sel_num", "syn_sel_num", "syn_long_spread", "syn_tie_pairs_max", "pf_list_fill", "tile_min_tuples", "tile_big_tiles_min", "rates"):
        assert name in hdr, name
