"""Extract the translation test vectors of the reference (seq/codon_tables_test.go:26-135: public GenBank
sequences with their expected translations) and the 64-letter NCBI amino-acid lines of its codon tables
(seq/codon_tables.go:431-621) into tests/golden/codon_golden.json.  Data only: inputs and expected outputs.

Run in the build container (needs /root/reference):  python tests/golden/make_codon_golden.py
"""
import json
import os
import re

REF = "/root/reference/seq"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    src = open(os.path.join(REF, "codon_tables_test.go")).read()
    vectors = []
    for m in re.finditer(r"codonTableTest\{(.*?)\n\t\t\}\)", src, re.S):
        body = m.group(1)
        table = int(re.search(r"table:\s*(\d+)", body).group(1))
        nt = re.sub(r"\s", "", re.search(r"nt:\s*re\.ReplaceAllString\(`([^`]*)`", body).group(1))
        aa = re.sub(r"\s", "", re.search(r"aa:\s*re\.ReplaceAllString\(`([^`]*)`", body).group(1))
        frame = int(re.search(r"frame:\s*(-?\d+)", body).group(1))
        trim = re.search(r"trim:\s*(\w+)", body).group(1) == "true"
        clean = re.search(r"clean:\s*(\w+)", body).group(1) == "true"
        vectors.append({"table": table, "frame": frame, "trim": trim, "clean": clean, "nt": nt, "aa": aa})
    tables = {}
    src = open(os.path.join(REF, "codon_tables.go")).read()
    for m in re.finditer(r"codonTableFromText\((\d+),\s*\"[^\"]*\",\s*`([^`\n]*)\n", src):
        tables[m.group(1)] = m.group(2)
    assert len(vectors) == 6 and len(tables) == 24, (len(vectors), len(tables))
    with open(os.path.join(HERE, "codon_golden.json"), "w") as f:
        json.dump({"source": "seq/codon_tables_test.go:26-135, seq/codon_tables.go:431-621", "vectors": vectors, "ncbieaa": tables}, f,
                  indent=1)
    print("wrote", len(vectors), "vectors,", len(tables), "tables")


if __name__ == "__main__":
    main()
