"""dev helper (CPU): random FASTA/Q text, valid and broken, through the library's reader and the reader oracle."""
import os, random, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bio_amd import fastx
from oracle import fastx_oracle as FO

def gen(rng):
    fq = rng.random() < 0.5
    nl = rng.choice(["\n", "\n", "\r\n"])
    out = []
    if rng.random() < 0.1: out.append("\n" * rng.randint(1, 3))
    for i in range(rng.randint(0, 8)):
        name = rng.choice(["r%d" % i, "r%d desc >x @y" % i, "", "@@", ">"])
        n = rng.choice([0, 1, 5, 17, 60])
        seq = "".join(rng.choice("ACGTN") for _ in range(n))
        width = rng.choice([4, 60, 1000])
        lines = [seq[j:j + width] for j in range(0, max(n, 1), width)]
        if fq:
            qual = "".join(rng.choice("@+>I#5") for _ in range(n if rng.random() < 0.9 else max(0, n + rng.choice([-1, 1, 3]))))
            qlines = [qual[j:j + width] for j in range(0, max(len(qual), 1), width)]
            rec = "@" + name + nl + nl.join(lines) + nl + ("+" + rng.choice(["", name]) + nl if rng.random() < 0.95 else "") + nl.join(qlines)
        else:
            rec = ">" + name + nl + nl.join(lines)
        if rng.random() < 0.9 or i < 7: rec += nl
        if rng.random() < 0.05: rec += nl
        out.append(rec)
    s = "".join(out)
    if rng.random() < 0.1 and s:
        p = rng.randrange(len(s)); s = s[:p] + rng.choice(["@", ">", "\n", "+", "x"]) + s[p:]
    return s.encode()

def read_lib(path):
    r = fastx.Reader(path); recs = []; err = None
    try:
        for c in r.chunks(rng_chunk[0]):
            for i in range(len(c)): recs.append((c.name(i), c.sequence(i), c.quality(i)))
    except fastx.FastxError as e: err = e
    r.Close(); return recs, err

def read_par(path, threads, piece):
    try:
        r = fastx.ParallelReader(path, threads, piece)
    except fastx.FastxError as e:
        return [], e
    seqs = []; err = None
    try:
        for seq, offs in r.pieces():
            seqs += [seq[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(len(offs) - 1)]
    except fastx.FastxError as e: err = e
    r.close(); return seqs, err

rng_chunk = [0]
first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
d = tempfile.mkdtemp()
for seed in range(first, first + count):
    rng = random.Random(seed)
    data = gen(rng)
    path = os.path.join(d, "x.fx")
    open(path, "wb").write(data)
    os.environ["BSK_FASTX_BUF"] = str(rng.choice([1, 2, 5, 16, 64, 4096]))
    rng_chunk[0] = rng.choice([0, 1, 3])
    want, _, oerr = FO.read_records(data)
    got, err = read_lib(path)
    ok = got == want and ((oerr is None) == (err is None))
    if ok and oerr is not None:
        ok = (isinstance(oerr, FO.NotFastx) and err is fastx.ErrNotFASTXFormat) or (not isinstance(oerr, FO.NotFastx) and err is fastx.ErrBadFASTQFormat)
    if ok:  # the block-parallel reader: the serial reader's sequences and error, whatever the piece size -- plain or BGZF
        piece, threads = rng.choice([1, 2, 3, 7, 20, 50, 200, 1 << 20]), rng.choice([1, 2, 4])
        if rng.random() < 0.4:
            path = os.path.join(d, "x.fx.gz")
            open(path, "wb").write(FO.bgzf_compress(data, rng.choice([1, 5, 40, 1000]), 1, rng))
        pgot, perr = read_par(path, threads, piece)
        ok = pgot == [s_ for _, s_, _ in want] and ((perr is None) == (err is None)) and (perr is None or perr is err or perr.code == err.code)
        if not ok: print("PARALLEL piece", piece, "threads", threads, pgot[:3], perr)
    if not ok:
        bad += 1
        print("SEED", seed, "buf", os.environ["BSK_FASTX_BUF"], repr(data[:200]), "\n  lib", got[:3], err, "\n  ora", want[:3], oerr)
        if bad >= 5: break
print("done", count, "cases,", bad, "failures")
