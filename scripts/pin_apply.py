#!/usr/bin/env python3
"""Apply the verdict of the Go pin run (bindings/go/pin/run.sh -> scripts/pin_diff.py --json pin_verdict.json) to the repository:
for every banner the upstream iterators CONFIRMED, the "parity unpinned" note is rewritten to "pinned by the upstream iterators
(bindings/go/pin, <date>)" in oracle/bio_oracle.h, include/biosketch.h and DESIGN.md, and the verdict is recorded under
tests/golden/pin_verdict.json (tests/test_pin_harness.py then checks that the banners and the record agree).  Nothing is touched for a
banner with a differing case -- pin_diff names the cases; fix the oracle (and the kernels) first.

    python scripts/pin_apply.py bindings/go/pin/pin_verdict.json [--dry-run]
"""
import datetime
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# banner -> [(file, regex of the unpinned note, replacement template)]
EDITS = {
    "wyhash": [("oracle/bio_oracle.h", r"(wyhash \(protein paths\) \.+ )PARITY UNPINNED", r"\1PINNED by upstream ({stamp})"),
               ("DESIGN.md", r"wyhash — no reference test checks a protein hash value,", "wyhash — pinned by the upstream iterators ({stamp}; before: no reference test checked a protein hash value),")],
    "first_window_tie": [("oracle/bio_oracle.h", r"(first-window tie order \.+ )PARITY UNPINNED", r"\1PINNED by upstream ({stamp})")],
    "non_acgt": [("oracle/bio_oracle.h", r"(non-ACGT bytes, k > 64 \.+ )PARITY UNPINNED", r"\1PINNED by upstream ({stamp}: non-ACGT bytes)")],
    "k_over_64": [],
    "syncmer": [("DESIGN.md", r"syncmers have no reference-held value at all", "syncmer values are pinned by the upstream iterators ({stamp}; the reference's own tests hold none)")],
}


def main(argv):
    dry = "--dry-run" in argv
    argv = [a for a in argv if a != "--dry-run"]
    if len(argv) != 2:
        print(__doc__)
        return 2
    v = json.load(open(argv[1]))
    stamp = "bindings/go/pin, " + datetime.date.today().isoformat()
    done = []
    for name, b in v["banners"].items():
        if not b.get("pinned"):
            print("banner %-17s kept (%d of %d cases differ)" % (name + ":", b.get("differing", 0), b.get("cases", 0)))
            continue
        for rel, pat, repl in EDITS.get(name, []):
            path = os.path.join(ROOT, rel)
            txt = open(path).read()
            new, n = re.subn(pat, repl.format(stamp=stamp), txt)
            print("banner %-17s %s: %d note(s) rewritten%s" % (name + ":", rel, n, " (dry run)" if dry else ""))
            if n and not dry:
                open(path, "w").write(new)
        done.append(name)
    if not dry:
        os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
        shutil.copyfile(argv[1], os.path.join(ROOT, "tests", "golden", "pin_verdict.json"))
    print("pinned banners:", ", ".join(done) if done else "none")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
