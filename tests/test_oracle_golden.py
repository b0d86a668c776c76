"""CPU: the oracle against the reference's known-answer test and the committed golden vectors."""
import json
import os
import random

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "sketches_golden.json")))


def test_reference_known_answer_pins_nthash(oracle):
    # sketches/sketch_test.go:33-76: the only value-checking test of the reference on this path
    kat = GOLD["ref_kat"]
    h, p, st, fl = oracle.minimizer(kat["seq"], kat["k"], kat["w"])
    assert h.tolist() == kat["codes"]
    assert p.tolist() == [0, 1, 4, 7, 8]
    # the same five values fall out of the plain hash stream at those positions
    hs, _ = oracle.nthash(kat["seq"], kat["k"])
    assert [int(hs[i]) for i in (0, 1, 4, 7, 8)] == kat["codes"]
    # closed form == state machine
    h2, p2, st2, fl2 = oracle.minimizer(kat["seq"], kat["k"], kat["w"], closed=True)
    assert h2.tolist() == kat["codes"] and p2.tolist() == p.tolist()


def test_reference_count_assertions(oracle):
    # iterator_test.go:63,100 ; iterator-protein_test.go is DNA-input (out of scope)
    s = "AAGTTTGAATCATTCAACTATCTAGTTTTCAGAGAACAATGTTCTCTAAAGAATAGAAAAGAGTCATTGTGCGGTGATGATGGCGGGAAGGATCCACCTG"
    assert len(oracle.kmer_codes(s, 10)) == len(s) - 10 + 1
    assert len(oracle.nthash(s, 10)[0]) == len(s) - 10 + 1
    for t in ("GAACAATGTTCTCTAAAATTG", "GcACAATGTTCTCTAAAATTG"):  # iterator_test.go:141
        assert len(oracle.simhash(t, 21, 5, 5)) == len(t) - 21 + 1
    # sketch_test.go:111 (commented-out expectation): 5 syncmers for k=5 s=2
    h, p, _, _ = oracle.syncmer("GGCAAGTTCGTCATCGATC", 5, 2)
    assert len(h) == 5


def _run(oracle, c):
    fn = c["fn"]
    s = c["seq"]
    try:
        if fn == "minimizer":
            r = oracle.minimizer(s, c["k"], c["w"])
        elif fn == "syncmer":
            r = oracle.syncmer(s, c["k"], c["s"])
        elif fn == "nthash":
            r = oracle.nthash(s, c["k"], c["canonical"], c["circular"])
        elif fn == "kmer":
            r = oracle.kmer_codes(s, c["k"], c["canonical"])
        elif fn == "simhash":
            r = oracle.simhash(s, c["k"], c["m"], c["scale"])
        elif fn == "protein_minimizer":
            r = oracle.protein_minimizer(s, c["k"], c["w"])
        elif fn == "protein_hashes":
            r = oracle.protein_hashes(s, c["k"])
        elif fn == "wyhash":
            r = [oracle.wyhash(s, c["seed"])]
        else:
            raise AssertionError(fn)
    except oracle.OracleError as e:
        return {"error": e.name}
    if isinstance(r, tuple):
        return [[int(v) for v in x] if hasattr(x, "__len__") else int(x) for x in r]
    return [int(v) for v in r]


def test_golden_vectors(oracle):
    assert len(GOLD["cases"]) > 200
    for c in GOLD["cases"]:
        assert _run(oracle, c) == c["out"], (c["name"], c["fn"], c.get("k"))


def test_state_machine_equals_closed_form(oracle):
    """sketch.go:205-477 restated line by line == leftmost-argmin closed form (what the kernels compute)."""
    rng = random.Random(7)
    for _ in range(60000):
        L = rng.randint(1, 220)
        s = "".join(rng.choice(rng.choice(["ACGT", "AC", "A", "ACGTN", "AAAC"])) for _ in range(L))
        k = rng.choice([1, 2, 3, 5, 7, 11, 21, 31, 33, 64, 70])
        w = rng.choice([1, 2, 3, 5, 11, 15, 20])
        circ = rng.random() < 0.2
        for fn, x in ((oracle.minimizer, w), (oracle.syncmer, rng.randint(1, k) if rng.random() < 0.9 else rng.choice([0, k + 1]))):
            res = []
            for closed in (False, True):
                try:
                    res.append(fn(s, k, x, circ, closed=closed))
                except oracle.OracleError as e:
                    res.append(e.name)
            a, b = res
            if isinstance(a, str) or isinstance(b, str):
                assert a == b, (s, k, x, circ)
            else:
                assert all(np.array_equal(a[i], b[i]) for i in range(3)) and a[3] == b[3], (s, k, x, circ)


def test_protein_state_machine_equals_closed_form(oracle):
    rng = random.Random(3)
    for _ in range(1500):
        L = rng.randint(1, 120)
        aa = "".join(rng.choice("ACDEFGHIKLMNPQRSTVWY"[: rng.choice([2, 20])]) for _ in range(L))
        k, w = rng.choice([1, 2, 3, 5, 9, 10]), rng.choice([1, 2, 3, 5, 7])
        res = []
        for closed in (False, True):
            try:
                res.append(oracle.protein_minimizer(aa, k, w, closed=closed))
            except oracle.OracleError as e:
                res.append(e.name)
        a, b = res
        if isinstance(a, str) or isinstance(b, str):
            assert a == b
        else:
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]


def test_nthash_is_the_published_formula(oracle):
    """rolling recurrence == from-scratch definition, incl. k > 64 (rotations mod 64)."""
    rng = random.Random(5)
    seed = {'A': 0x3c8bfbb395c60474, 'C': 0x3193c18562a02b4c, 'G': 0x20323ed082572324, 'T': 0x295549f54be24456}
    comp = {'A': 'T', 'C': 'G', 'G': 'C', 'T': 'A'}

    def rol(v, n):
        n %= 64
        return ((v << n) | (v >> (64 - n))) & (2 ** 64 - 1) if n else v
    x = "".join(rng.choice("ACGT") for _ in range(120))
    for k in (1, 5, 21, 31, 64, 65, 100):
        hs, st = oracle.nthash(x, k)
        for i in range(len(x) - k + 1):
            f = r = 0
            for j in range(k):
                f ^= rol(seed[x[i + j]], k - 1 - j)
                r ^= rol(seed[comp[x[i + j]]], j)
            assert int(hs[i]) == min(f, r) and int(st[i]) == (1 if r < f else 0)


def test_error_surface(oracle):
    with pytest.raises(oracle.OracleError) as e:
        oracle.minimizer("ACGT", 0, 3)
    assert e.value.name == "ErrInvalidK"
    with pytest.raises(oracle.OracleError) as e:
        oracle.minimizer("ACGT", 3, 0)
    assert e.value.name == "ErrInvalidW"
    with pytest.raises(oracle.OracleError) as e:
        oracle.minimizer("ACGTACG", 5, 4)  # len 7 < k+w-1 = 8
    assert e.value.name == "ErrShortSeq"
    assert len(oracle.minimizer("ACGTACGT", 5, 4)[0]) >= 1
    with pytest.raises(oracle.OracleError) as e:
        oracle.syncmer("ACGTACGTAC", 5, 6)
    assert e.value.name == "ErrInvalidS"
    with pytest.raises(oracle.OracleError) as e:
        oracle.syncmer("ACGTACGTAC", 5, 0)
    assert e.value.name == "ErrInvalidS"
    with pytest.raises(oracle.OracleError) as e:
        oracle.simhash("ACGTACGTACGT", 8, 3, 1)
    assert e.value.name == "ErrInvalidM"
    with pytest.raises(oracle.OracleError) as e:
        oracle.simhash("ACGTACGTACGT", 8, 5, 5)
    assert e.value.name == "ErrInvalidScale"
    with pytest.raises(oracle.OracleError) as e:
        oracle.kmer_codes("ACGTXACGT", 4)  # N maps to A in kmers.go:23-40; X is illegal
    assert e.value.name == "ErrIllegalBase"
    with pytest.raises(oracle.OracleError) as e:
        oracle.protein_minimizer("ACDEFGHIKLMNPQRSTVWY", 7, 3)  # len 20 < 3k
    assert e.value.name == "ErrShortSeq"


def _codon_golden():
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "codon_golden.json")))


def test_translation_oracle_is_pinned_by_the_reference_vectors(oracle):
    """seq/codon_tables_test.go:26-135: six GenBank sequences with expected translations (frames 1,-1,-2,-3; trim;
    an N codon), and the 64-letter NCBI amino-acid line of all 24 genetic codes (seq/codon_tables.go:431-621)."""
    g = _codon_golden()
    assert len(g["vectors"]) == 6 and len(g["ncbieaa"]) == 24
    for tid, aa in g["ncbieaa"].items():
        assert oracle.genetic_code(int(tid)) == aa, tid
    for v in g["vectors"]:
        assert oracle.translate(v["nt"], v["table"], v["frame"], v["trim"], v["clean"]) == v["aa"], (v["table"], v["frame"])
    with pytest.raises(ValueError):
        oracle.translate("ACGTACGT", 7, 1)
    with pytest.raises(ValueError):
        oracle.translate("ACGTACGT", 1, 0)
    with pytest.raises(ValueError):
        oracle.translate("AC", 1, 1)


def test_translation_quirks_of_the_reference(oracle):
    # ambiguity letters resolve only when every base they stand for agrees (codon_tables.go:350-427)
    assert oracle.translate("GCNACNCTNRAYAARTTYCGNMGN", 1, 1) == "ATLXKFRX"   # RAY = N|D -> X ; MGN = CGN (R) | AGN (R|S) -> X
    assert oracle.translate("ATGRAYTGGNNNGCN---TAA", 1, 1) == "MXWXA-*"   # "---" -> '-' (codon_tables.go:167)
    assert oracle.translate("AT-ATGA*A", 1, 1) == "XMX"                     # gap letters have the empty set -> X
    assert oracle.translate("ATGJJJTAA", 1, 1) == "MX*"                     # a letter outside base2code -> X (allowUnknownCodon)
    assert oracle.translate("augUAA", 1, 1) == "M*"                         # RNA, lower case
    # minus frames complement acgtACGT only (DNA.PairLetter, alphabet.go:353-359): 'R' stays 'R', 'U' stays 'U'
    assert oracle.translate("TTACAT", 1, -1) == "M*"
    assert oracle.translate("YTACAT", 1, -1) == "MY"    # a true complement (R) would read TAR = stop; the kept Y reads TAY = Y
    assert oracle.translate("UUACAU", 1, -1) == "LF"    # U is kept: codons "UTG" (= TTG, L) and "TUU" (= TTT, F), not M*
    # length of a translation: floor((L - f + 1) / 3) for frame f, same for -f
    for L in range(3, 40):
        for f in (1, 2, 3):
            s = "ACG" * 14
            assert len(oracle.translate(s[:L], 1, f)) == (L - f + 1) // 3 == len(oracle.translate(s[:L], 1, -f))


def test_protein_paths_on_nucleotides_check_the_input_length(oracle):
    k, w = 4, 3
    dna = "ATGGCCATTGTAATGGGCCGCTGAAAGGGTGCCCGATAG"
    aa = oracle.translate(dna, 1, 1)
    # k-mer hashes of the translation are the protein iterator's output
    assert np.array_equal(oracle.protein_hashes_nt(dna, k, 1, 1), np.array([oracle.wyhash(aa[i:i + k].encode()) for i in range(len(aa) - k + 1)], np.uint64))
    with pytest.raises(oracle.OracleError) as e:
        oracle.protein_hashes_nt(dna[:3 * k - 1], k, 1, 1)       # iterator-protein.go:50
    assert e.value.name == "ErrShortSeq"
    assert len(oracle.protein_hashes_nt(dna[:3 * k], k, 1, 2)) == 0   # 12 nt pass the check; frame 2 leaves 3 residues < k: nothing, no error
    with pytest.raises(oracle.OracleError):
        oracle.protein_minimizer_nt(dna[:3 * k + w - 2], k, w, 1, 1)    # sketch-protein.go:73
    h, p, _ = oracle.protein_minimizer_nt(dna[:3 * k + w - 1], k, w, 1, 1)   # 14 nt -> 4 residues -> 1 k-mer < w: no window completes
    assert len(h) == 0 and len(p) == 0
    h, p, _ = oracle.protein_minimizer_nt(dna, k, w, 1, 1)
    h2, p2, _ = oracle.protein_minimizer(aa + "A" * 20, k, w)             # same machine on protein input (length padded past 3k+w-1)
    assert len(h) > 0 and np.array_equal(p, p2[:len(p)]) and np.array_equal(h, h2[:len(h)])


def test_reference_protein_tests_feed_dna(oracle):
    """TestProteinIterator (iterator-protein_test.go:29-63) and TestProteinMinimizer (sketch-protein_test.go:29-58) build
    the iterators from a DNA Seq; the only assertion is the count len/3 - k + 1."""
    dna = "AAGTTTGAATCATTCAACTATCTAGTTTTCAGAGAACAATGTTCTCTAAAGAATAGAAAAGAGTCATTGTGCGGTGATGATGGCGGGAAGGATCCACCTG"
    k = 10
    assert len(oracle.protein_hashes_nt(dna, k, 1, 1)) == len(dna) // 3 - k + 1
    h, p, _ = oracle.protein_minimizer_nt(dna, k, 3, 1, 1)
    assert 1 <= len(h) <= len(dna) // 3 - k + 1 and list(p) == sorted(set(int(x) for x in p))


def test_two_strand_kmer_codes_pair_letters_by_alphabet_hand_derived(oracle):
    """NextKmer, canonical=false (iterator.go:713-723): the second strand is RevComInplace of the Seq, whose Alphabet decides which
    letters pair (seq/alphabet.go:353-383) -- worked out by hand for ACGUUNRT, k = 3 (A0 C1 G2 T/U3; N, R -> 0; Y -> 1):
      forward ACG CGU GUU UUN UNR NRT = 6 27 47 60 48 3 for every alphabet;
      RNA          : T, R stay, N -> N, U -> A, G <-> C, A -> U : TRNAACGU -> TRN RNA NAA AAC ACG CGU = 48 0 0 1 6 27
      DNAredundant : T -> A, R -> Y, U stays               : AYNUUCGT -> AYN YNU NUU UUC UCG CGT = 4 19 15 61 54 27
      Unlimit      : reversed, not complemented            : TRNUUGCA -> TRN RNU NUU UUG UGC GCA = 48 3 15 62 57 36"""
    fwd = [6, 27, 47, 60, 48, 3]
    for alphabet, second in ((3, [48, 0, 0, 1, 6, 27]), (0, [4, 19, 15, 61, 54, 27]), (5, [48, 3, 15, 62, 57, 36])):
        assert list(oracle.kmer_codes("ACGUUNRT", 3, False, False, alphabet)) == fwd + second, alphabet
    # plain DNA pairs acgt only (T -> A, R / N / U stay): ARNUUCGT; RNAredundant: T stays, R -> Y, U -> A: TYNAACGU
    assert list(oracle.kmer_codes("ACGUUNRT", 3, False, False, 2))[6:] == [0, 3, 15, 61, 54, 27]
    assert list(oracle.kmer_codes("ACGUUNRT", 3, False, False, 4))[6:] == [52, 16, 0, 1, 6, 27]
