#!/usr/bin/env python3
"""Compare the output of the Go pin harness (bindings/go/pin/pin_test.go: the REAL shenwei356/bio iterators) with the committed
goldens (tests/golden/sketches_golden.json: the CPU oracle's values) and say which "parity unpinned" banners can be removed.

    cd bindings/go/pin && go test -tags pin -run TestPin -v        # writes pin_out.jsonl (needs a Go toolchain; this image has none)
    python scripts/pin_diff.py bindings/go/pin/pin_out.jsonl

Only what the upstream API exposes is compared: the values of Next*() and Index() (and the error a constructor / NextKmer returns);
the engine's strand bits and status flags have no upstream counterpart.  Exit status 0 iff every case agrees.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "sketches_golden.json")
TIE = 0x10  # BSK_ST_FIRST_WINDOW_TIE

# banner -> (where it is written, which cases speak to it)
BANNERS = {
    "wyhash": ("oracle/bio_oracle.h, include/biosketch.h, DESIGN.md 1: wyhash v1 restated (rows A7/A8)",
               lambda c: c["fn"] in ("wyhash", "protein_hashes", "protein_minimizer")),
    "syncmer": ("DESIGN.md 1: no reference-held syncmer value", lambda c: c["fn"] == "syncmer"),
    "first_window_tie": ("BSK_ST_FIRST_WINDOW_TIE: order of equal hashes after sorts.Quicksort",
                         lambda c: c["fn"] in ("minimizer", "syncmer", "protein_minimizer") and isinstance(c["out"], list) and (c["out"][-1] & TIE)),
    "non_acgt": ("BSK_ST_HAS_NON_ACGT: ntHash seeds of bytes outside ACGTacgt",
                 lambda c: c["fn"] in ("nthash", "minimizer", "syncmer", "simhash") and any(ch not in "ACGTacgt" for ch in c["seq"])),
    "k_over_64": ("DESIGN.md 1: rotation modulo 64 for k > 64", lambda c: c.get("k", 0) > 64 and c["fn"] in ("nthash", "minimizer", "syncmer")),
}


def golden_view(c):
    """(values, index or None, error or None) of a golden case, as the upstream API would show it"""
    out = c["out"]
    if isinstance(out, dict):
        return None, None, out["error"]
    fn = c["fn"]
    if fn in ("minimizer", "syncmer", "protein_minimizer"):
        return out[0], out[1], None
    if fn == "nthash":
        return out[0], None, None
    return out, None, None


def key(c):
    return (c["name"], c["fn"], c.get("k"), c.get("w"), c.get("s"), c.get("m"), c.get("scale"), c.get("seed"), c.get("canonical"), c.get("circular", False))


def diff(golden_cases, pinned):
    """-> list of (case, what differs); pinned: list of harness records in the order of the golden cases"""
    bad = []
    if len(pinned) != len(golden_cases):
        return [(None, "the harness wrote %d records for %d golden cases" % (len(pinned), len(golden_cases)))]
    for c, p in zip(golden_cases, pinned):
        if p.get("name") != c["name"] or p.get("fn") != c["fn"]:
            bad.append((c, "record order: %s/%s" % (p.get("name"), p.get("fn"))))
            continue
        vals, idx, err = golden_view(c)
        perr = p.get("error") or None
        if err is not None or (perr is not None and c["fn"] != "kmer"):
            if err != perr:
                bad.append((c, "error %r, upstream %r" % (err, perr)))
            continue
        if c["fn"] == "kmer" and perr is not None:  # NextKmer stops at ErrIllegalBase: the codes before it stay
            bad.append((c, "upstream stopped with %s after %d codes, golden has %d" % (perr, len(p["values"]), len(vals))))
            continue
        if [int(v) for v in p["values"]] != [int(v) for v in vals]:
            n = min(len(vals), len(p["values"]))
            first = next((i for i in range(n) if int(vals[i]) != int(p["values"][i])), n)
            bad.append((c, "values differ from entry %d (%d golden, %d upstream)" % (first, len(vals), len(p["values"]))))
        elif idx is not None and [int(v) for v in (p.get("index") or [])] != [int(v) for v in idx]:
            bad.append((c, "Index() differs"))
    return bad


def main(argv):
    out_json = None
    if "--json" in argv:  # machine-readable verdict for scripts/pin_apply.py
        i = argv.index("--json")
        out_json = argv[i + 1]
        argv = argv[:i] + argv[i + 2:]
    if len(argv) != 2:
        print(__doc__)
        return 2
    golden_cases = json.load(open(GOLDEN))["cases"]
    pinned = [json.loads(line) for line in open(argv[1]) if line.strip()]
    bad = diff(golden_cases, pinned)
    bad_ids = {id(c) for c, _ in bad if c is not None}
    for c, what in bad:
        print("MISMATCH %s: %s" % ("%s/%s k=%s" % (c["name"], c["fn"], c.get("k")) if c else "-", what))
    print("%d of %d cases agree with upstream" % (len(golden_cases) - len(bad), len(golden_cases)))
    verdicts = {}
    for name, (where, pred) in BANNERS.items():
        cases = [c for c in golden_cases if pred(c)]
        wrong = [c for c in cases if id(c) in bad_ids]
        if not cases:
            verdict = "no case speaks to it"
        elif wrong:
            verdict = "KEEP (%d of %d cases differ from upstream: fix the oracle first)" % (len(wrong), len(cases))
        else:
            verdict = "can be REMOVED (%d cases pinned by upstream)" % len(cases)
        verdicts[name] = dict(cases=len(cases), differing=len(wrong), pinned=bool(cases) and not wrong, where=where)
        print("banner %-17s %s -- %s" % (name + ":", verdict, where))
    if out_json:
        json.dump(dict(cases=len(golden_cases), agreeing=len(golden_cases) - len(bad), banners=verdicts,
                       mismatches=[dict(name=c["name"], fn=c["fn"], k=c.get("k"), what=what) for c, what in bad if c]), open(out_json, "w"), indent=1)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
