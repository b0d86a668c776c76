"""The Go pin harness (bindings/go/pin) and scripts/pin_diff.py: the route to pinning wyhash, syncmers and first-window ties
against the real Go iterators.  No Go toolchain here, so: the harness source must handle every `fn` of the golden file and call
the upstream constructors with the upstream signatures; pin_diff must accept an upstream that agrees and catch one that does not."""
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = json.load(open(os.path.join(HERE, "golden", "sketches_golden.json")))
SRC = open(os.path.join(ROOT, "bindings", "go", "pin", "pin_test.go")).read()
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import pin_diff  # noqa: E402


def test_harness_enumerates_every_fn_of_the_golden_file():
    fns = {c["fn"] for c in GOLD["cases"]}
    handled = set(re.findall(r'case "(\w+)":', SRC))
    assert fns <= handled, fns - handled
    assert SRC.startswith("//go:build pin")
    # the upstream constructors, with the argument order of sketches/*.go (iterator.go:113,615,668; sketch.go:85,142;
    # iterator-protein.go:46; sketch-protein.go:62)
    for call in ("sketches.NewMinimizerSketch(s, c.K, c.W, c.Circular)", "sketches.NewSyncmerSketch(s, c.K, c.S, c.Circular)",
                 "sketches.NewHashIterator(s, c.K, canonical(c), c.Circular)", "sketches.NewKmerIterator(s, c.K, canonical(c), c.Circular)",
                 "sketches.NewSimHashIterator(s, c.K, c.M, c.Scale, canonical(c), c.Circular)", "sketches.NewProteinIterator(s, c.K, 1, 1)",
                 "sketches.NewProteinMinimizerSketch(s, c.K, 1, 1, c.W)", "wyhash.Hash([]byte(c.Seq), c.Seed)"):
        assert call in SRC, call
    # every parameter a golden case carries is a field the harness reads
    keys = set().union(*(set(c) for c in GOLD["cases"])) - {"out"}
    tags = set(re.findall(r'json:"(\w+)"', SRC))
    assert keys <= tags, keys - tags


def _as_upstream(c):
    vals, idx, err = pin_diff.golden_view(c)
    r = {"name": c["name"], "fn": c["fn"], "values": vals or []}
    if idx is not None:
        r["index"] = idx
    if err:
        r["error"] = err
    return r


def test_pin_diff_accepts_agreement_and_catches_a_difference(tmp_path):
    cases = GOLD["cases"]
    assert {"wyhash", "syncmer", "protein_hashes"} <= {c["fn"] for c in cases}
    assert any(pred(c) for c in cases for _, pred in [pin_diff.BANNERS["first_window_tie"]])
    recs = [_as_upstream(c) for c in cases]
    assert pin_diff.diff(cases, recs) == []
    out = tmp_path / "pin_out.jsonl"
    out.write_text("".join(json.dumps(r) + "\n" for r in recs))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pin_diff.py"), str(out)], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.count("can be REMOVED") == len(pin_diff.BANNERS), p.stdout
    # an upstream whose wyhash differs in one value
    i = next(i for i, c in enumerate(cases) if c["fn"] == "protein_hashes")
    recs[i] = dict(recs[i], values=[recs[i]["values"][0] ^ 1] + recs[i]["values"][1:])
    bad = pin_diff.diff(cases, recs)
    assert len(bad) == 1 and bad[0][0] is cases[i]
    out.write_text("".join(json.dumps(r) + "\n" for r in recs))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pin_diff.py"), str(out)], capture_output=True, text=True)
    assert p.returncode == 1 and "banner wyhash:" in p.stdout and "KEEP" in p.stdout, p.stdout


def test_pin_apply_rewrites_exactly_the_confirmed_banners(tmp_path):
    """scripts/pin_apply.py (the last step of bindings/go/pin/run.sh --apply): with an agreeing upstream every note it knows is found in
    the tree (dry run: nothing is written), with a differing wyhash the wyhash note is kept."""
    cases = GOLD["cases"]
    recs = [_as_upstream(c) for c in cases]
    out = tmp_path / "pin_out.jsonl"
    out.write_text("".join(json.dumps(r) + "\n" for r in recs))
    verdict = tmp_path / "pin_verdict.json"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pin_diff.py"), str(out), "--json", str(verdict)], capture_output=True, text=True)
    assert p.returncode == 0 and json.load(open(verdict))["banners"]["wyhash"]["pinned"]
    before = {f: open(os.path.join(ROOT, f)).read() for f in ("oracle/bio_oracle.h", "DESIGN.md")}
    q = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pin_apply.py"), str(verdict), "--dry-run"], capture_output=True, text=True)
    assert q.returncode == 0, q.stdout + q.stderr
    assert "0 note(s) rewritten" not in q.stdout, q.stdout  # every note pin_apply knows still exists in the tree
    assert "pinned banners: wyhash" in q.stdout.replace("\n", " ") or "wyhash" in q.stdout.splitlines()[-1]
    assert before == {f: open(os.path.join(ROOT, f)).read() for f in before}  # a dry run writes nothing
    i = next(i for i, c in enumerate(cases) if c["fn"] == "protein_hashes")
    recs[i] = dict(recs[i], values=[recs[i]["values"][0] ^ 1] + recs[i]["values"][1:])
    out.write_text("".join(json.dumps(r) + "\n" for r in recs))
    subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pin_diff.py"), str(out), "--json", str(verdict)], capture_output=True, text=True)
    q = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pin_apply.py"), str(verdict), "--dry-run"], capture_output=True, text=True)
    assert "banner wyhash:" in q.stdout and "kept" in [ln for ln in q.stdout.splitlines() if ln.startswith("banner wyhash:")][0]
    # the one-command wrapper names both tools
    sh = open(os.path.join(ROOT, "bindings", "go", "pin", "run.sh")).read()
    assert "pin_diff.py" in sh and "pin_apply.py" in sh and "go test -tags pin -run TestPin" in sh
