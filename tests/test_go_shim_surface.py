"""The Go shim (bindings/go/sketches) must present the reference package's exported surface: every exported func, method, type
and sentinel of /root/reference/sketches/*.go with the same signature (parameter and result TYPES; names are free).

The reference is parsed as text at test time in the build container (it does not travel to the GPU box: the test skips there).
No Go toolchain exists in this image, so this is the strongest check available for the shim: names + signatures + the ABI
call sequences (tests/cpp/test_sketches.cpp replays the latter on the GPU).
"""
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/sketches"
SHIM = os.path.join(ROOT, "bindings", "go", "sketches")


def _types(params: str):
    """'s *seq.Seq, k int, m, n int' -> ['*seq.Seq', 'int', 'int', 'int'];  '(code uint64, ok bool)' -> ['uint64', 'bool']"""
    params = params.strip()
    if params.startswith("(") and params.endswith(")"):
        params = params[1:-1]
    if not params.strip():
        return []
    parts = [p.strip() for p in params.split(",")]
    out, pending = [], 0
    named = any(len(p.split()) >= 2 for p in parts)
    for p in parts:
        toks = p.split()
        if named:
            if len(toks) == 1:  # a name whose type follows later ("k, w int")
                pending += 1
            else:
                out.extend([" ".join(toks[1:])] * (pending + 1))
                pending = 0
        else:
            out.append(p)
    assert pending == 0, params
    return out


_FUNC = re.compile(r"^func\s+(?:\((\w+)\s+(\*?\w+)\)\s+)?(\w+)\((.*?)\)\s*(\(.*?\)|[\w\*\.\[\]]+)?\s*\{", re.M)
_TYPE = re.compile(r"^type\s+([A-Z]\w*)\s+(struct|interface|\w+)", re.M)
_VAR = re.compile(r"^(?:var\s+|\t)(Err\w+)\s*=\s*(?:fmt\.Errorf|errors\.New)\(\"([^\"]*)\"\)", re.M)


def surface(files):
    funcs, types, errs, fields = {}, set(), {}, {}
    for f in files:
        src = open(f).read()
        for m in _FUNC.finditer(src):
            recv, name = m.group(2), m.group(3)
            if not name[0].isupper():
                continue
            if recv and not recv.lstrip("*")[0].isupper():
                continue
            key = (recv.lstrip("*") + "." if recv else "") + name
            funcs[key] = (("*" if recv and recv.startswith("*") else "") if recv else "", tuple(_types(m.group(4))), tuple(_types(m.group(5) or "")))
        for m in _TYPE.finditer(src):
            types.add(m.group(1))
        for m in _VAR.finditer(src):
            errs[m.group(1)] = m.group(2)
        m = re.search(r"type IdxValue struct \{(.*?)\}", src, re.S)
        if m:
            fields["IdxValue"] = [tuple(ln.split("//")[0].split()) for ln in m.group(1).strip().splitlines() if ln.strip()]
    return funcs, types, errs, fields


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")
def test_shim_exports_the_reference_surface():
    ref_files = [f for f in glob.glob(os.path.join(REF, "*.go")) if not f.endswith("_test.go")]
    shim_files = glob.glob(os.path.join(SHIM, "*.go"))
    rf, rt, re_, rfields = surface(ref_files)
    sf, st, se, sfields = surface(shim_files)
    assert len(rf) >= 20 and {"Iterator", "Sketch", "ProteinIterator", "ProteinMinimizerSketch", "IdxValue"} <= rt and len(re_) == 11
    missing = sorted(set(rf) - set(sf))
    assert not missing, f"functions/methods of the reference missing in the shim: {missing}"
    diff = {k: (rf[k], sf[k]) for k in rf if rf[k] != sf[k]}
    assert not diff, f"signatures differ (receiver kind, parameter types, result types): {diff}"
    assert rt <= st, f"exported types missing: {sorted(rt - st)}"
    assert re_ == {k: se.get(k) for k in re_}, "sentinel errors: names and texts must be the reference's"
    assert rfields["IdxValue"] == sfields["IdxValue"]
    # and nothing the reference lacks may hide behind a reference TYPE (e.g. an embedded Iterator giving ProteinIterator a NextHash)
    for t in ("ProteinIterator", "ProteinMinimizerSketch", "Iterator"):
        extra = sorted(k for k in sf if k.startswith(t + ".") and k not in rf)
        assert not extra, f"{t} has exported methods upstream does not: {extra}"
    for f in shim_files:  # no embedding of the cursor types
        assert not re.search(r"struct\s*\{\s*(Iterator|Sketch)\s*\}", open(f).read()), f


def test_shim_maps_every_abi_error_code_and_only_calls_declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "biosketch.h")).read()
    declared = set(re.findall(r"\b(bsk_\w+)\s*\(", hdr))
    src = "".join(open(f).read() for f in glob.glob(os.path.join(SHIM, "*.go")))
    called = set(re.findall(r"C\.(bsk_\w+)\(", src))
    assert called and called <= declared, sorted(called - declared)
    for code in ("INVALID_K", "EMPTY_SEQ", "SHORT_SEQ", "ILLEGAL_BASE", "K_TOO_LARGE", "INVALID_M", "INVALID_SCALE", "INVALID_S", "INVALID_W",
                 "BUF_NIL", "BUF_NOT_EMPTY"):
        assert f"C.BSK_ERR_{code}:" in src and f"BSK_ERR_{code}" in hdr


# header entries the Go shim deliberately leaves unbound -- each with its reason; anything else in include/biosketch.h must be called from Go
GO_UNBOUND = {
    "bsk_batch_synth": "synthetic batches: bench / tests only",
    "bsk_sketch_timed": "HIP-event timing of repeated launches: bench only",
    "bsk_build_has_experiments": "make EXPERIMENTS=1 probe: tests only",
    "bsk_ctx_reload_options": "developer switches (BSK_*): tests only",
    "bsk_codon_lut": "host copy of the device codon table: tests only (the reference has its own seq.CodonTables)",
    "bsk_batch_fetch_ascii": "reads a batch back: tests only",
    "bsk_pipeline_run": "C callback form of the consumer loop: a Go host loops Pipeline.Next (no callback across cgo)",
    # the reader: a Go host HAS seqio/fastx; the C reader exists for hosts without one and feeds bsk_pipeline_open_fastx internally
    "bsk_fastx_open": "Go hosts read with seqio/fastx", "bsk_fastx_read_chunk": "Go hosts read with seqio/fastx", "bsk_fastx_info": "Go hosts read with seqio/fastx",
    "bsk_fastx_error": "Go hosts read with seqio/fastx", "bsk_fastx_close": "Go hosts read with seqio/fastx", "bsk_fastx_par_open": "Go hosts read with seqio/fastx",
    "bsk_fastx_par_next": "Go hosts read with seqio/fastx", "bsk_fastx_piece_data": "Go hosts read with seqio/fastx",
    "bsk_fastx_piece_release": "Go hosts read with seqio/fastx", "bsk_fastx_par_info": "Go hosts read with seqio/fastx",
    "bsk_fastx_par_error": "Go hosts read with seqio/fastx", "bsk_fastx_par_close": "Go hosts read with seqio/fastx",
    "bsk_batch_from_fastx": "Go hosts read with seqio/fastx",
}


def test_every_header_entry_has_a_go_binding():
    """include/biosketch.h and the shim move together: an entry added to the header fails this test until engine.go / pipeline.go /
    device.go call it (or it is listed above with its reason)."""
    hdr = open(os.path.join(ROOT, "include", "biosketch.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)  # prototypes only, not the prose
    declared = set(re.findall(r"\b(bsk_\w+)\s*\(", hdr))
    src = "".join(open(f).read() for f in glob.glob(os.path.join(SHIM, "*.go")))
    called = set(re.findall(r"C\.(bsk_\w+)\(", src))
    assert len(declared) >= 70
    unbound = sorted(declared - called - set(GO_UNBOUND))
    assert not unbound, f"header entries without a Go binding (bind them or list them in GO_UNBOUND with a reason): {unbound}"
    stale = sorted(set(GO_UNBOUND) & called) + sorted(set(GO_UNBOUND) - declared)
    assert not stale, f"GO_UNBOUND lists entries that are bound or no longer declared: {stale}"
    # struct fields the shim reads must exist in the header (a renamed field is a silent cgo build break we cannot see here)
    for struct, fields in (("bsk_chunk", re.findall(r"\bc\.(\w+)", src)), ("bsk_pipeline_stats", re.findall(r"\bst\.(\w+)", src))):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), hdr, re.S).group(1)
        for f in set(fields):
            assert re.search(r"\b%s\b" % f, body), (struct, f)
