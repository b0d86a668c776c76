// fastx.cpp -- host-side FASTA/FASTQ record reader that feeds bsk_batch_from_ascii: the packer in
// front of the sketch path.  Pure host code, part of libbiosketch.so.
//
// Record semantics follow seqio/fastx/reader.go (Reader.Read :233-369, parseRecord :372-471):
//   * the format is decided by the first byte that is not '\n': '>' FASTA, '@' FASTQ, anything else is not FASTA/Q
//     (:273-305); a file of newlines only, or an empty file, has no records;
//   * a record starts at a delimiter that follows '\n' -- a '>' or '@' inside a line is data (:312-352);
//   * lines lose one trailing '\r' (dropCR); FASTA: the header is the first line, the sequence the concatenation of the
//     others (:381-393); FASTQ: sequence lines up to the first non-empty line starting with '+', quality lines after it
//     (:394-423), so multi-line FASTQ works;
//   * a quality line may start with '@': when the bytes collected so far have a shorter quality than sequence the
//     candidate delimiter was data and reading continues (:326-335); a longer quality is ErrBadFASTQFormat;
//   * a record whose first line is empty keeps everything as its header and has no sequence (:425-429); a record with
//     neither header nor sequence ends the file (:440-442);
//   * the alphabet is guessed from (the first 10 000 letters of) the first sequence (:432-438, seq/alphabet.go:413-452).
// gzip input is read through zlib (gzread also passes plain files through), "-" is stdin.
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "biosketch.h"

struct bsk_ctx;
struct bsk_batch;

struct bsk_fastx {
    gzFile fh = nullptr;
    std::vector<uint8_t> buf;  // file window
    size_t r = 0, n = 0;       // unread part of buf: [r, n)
    bool eof = false, started = false, finished = false, io_error = false;
    int is_fastq = -1;
    int pending = 0;  // error met after some records of a chunk were read: returned by the next call
    int alphabet = -2;  // -2 not guessed yet, -1 "Unlimit", else BSK_ALPHA_*
    uint8_t delim = 0;
    std::string rec;  // bytes of the record being collected (after its delimiter)
    std::string err;
    // chunk storage handed to the caller
    std::vector<uint8_t> seq, name, qual;
    std::vector<uint64_t> seq_off, name_off;
    // scratch of parse()
    std::string p_head, p_seq, p_qual;
};

namespace {

bool fill(bsk_fastx *f) {  // refill the window; false at end of file
    if (f->eof) return false;
    f->r = 0;
    const int got = gzread(f->fh, f->buf.data(), (unsigned)f->buf.size());
    if (got <= 0) {
        // 0 is the end of the file; a damaged or truncated gzip stream / an I/O error shows as -1, or as 0 with a pending
        // Z_BUF_ERROR.  The reference reader hands a read error to the caller and yields nothing more (reader.go:262-268).
        int zerr = Z_OK;
        const char *msg = gzerror(f->fh, &zerr);
        if (got < 0 || (zerr != Z_OK && zerr != Z_STREAM_END)) {
            f->io_error = true;
            f->err = std::string("fastx: read error: ") + (msg && *msg ? msg : "unknown");
        }
        f->eof = true;
        f->n = 0;
        return false;
    }
    f->n = (size_t)got;
    return true;
}

inline void append_line(std::string &dst, const char *b, const char *e) {  // dropCR
    if (e > b && e[-1] == '\r') --e;
    dst.append(b, e);
}

// parseRecord: 0 ok, 1 sequence longer than quality, 2 quality longer, 3 nothing (end of file)
int parse(bsk_fastx *f) {
    const std::string &p = f->rec;
    f->p_head.clear();
    f->p_seq.clear();
    f->p_qual.clear();
    const char *b = p.data(), *e = b + p.size();
    const char *nl = (const char *)memchr(b, '\n', p.size());
    if (nl && nl > b) {
        append_line(f->p_head, b, nl);
        const char *q = nl + 1;
        bool is_qual = false;
        while (q <= e) {
            const char *l = q < e ? (const char *)memchr(q, '\n', (size_t)(e - q)) : nullptr;
            const char *le = l ? l : e;
            if (!f->is_fastq) append_line(f->p_seq, q, le);
            else if (l && le > q && *q == '+' && !is_qual) is_qual = true;  // reader.go:398
            else if (is_qual) append_line(f->p_qual, q, le);
            else if (l) append_line(f->p_seq, q, le);  // a last sequence line without '\n' is dropped there too (:407-411)
            if (!l) break;
            q = l + 1;
        }
        if (f->is_fastq && f->p_seq.size() != f->p_qual.size()) return f->p_seq.size() > f->p_qual.size() ? 1 : 2;
    } else {  // reader.go:425-429: no header line -- everything is the header
        const char *he = e;
        if (he > b && he[-1] == '\n') --he;
        append_line(f->p_head, b, he);
    }
    if (f->p_head.empty() && f->p_seq.empty()) return 3;
    return 0;
}

int guess_alphabet(const std::string &s) {  // GuessAlphabetLessConservatively, seq/alphabet.go:413-452
    if (s.empty()) return -1;
    bool seen[256] = {false};
    const size_t lim = s.size() < 10000 ? s.size() : 10000;
    for (size_t i = 0; i < lim; ++i) seen[(uint8_t)s[i]] = true;
    auto subset = [&](const char *letters) {
        bool ok[256] = {false};
        for (const char *c = letters; *c; ++c) ok[(uint8_t)*c] = true;
        for (int i = 0; i < 256; ++i)
            if (seen[i] && !ok[i]) return false;
        return true;
    };
    // the reference's order (seq/alphabet.go:423-437): DNA, RNA, DNAredundant, RNAredundant, Protein; letters + gaps + ambiguous of :353-383
    if (subset("acgtACGT -.nN")) return BSK_ALPHA_DNA_PLAIN;
    if (subset("acguACGU -.nN")) return BSK_ALPHA_RNA;
    if (subset("acgtryswkmbdhvACGTRYSWKMBDHV -.nN")) return BSK_ALPHA_DNA;
    if (subset("acguryswkmbdhvACGURYSWKMBDHV -.nN")) return BSK_ALPHA_RNA_REDUNDANT;
    if (subset("abcdefghijklmnopqrstuvwyzABCDEFGHIJKLMNOPQRSTUVWYZ -xX*_.")) return BSK_ALPHA_PROTEIN;
    return -1;
}

// next record into p_head / p_seq / p_qual: 1 record, 0 end of file, <0 error
int next_record(bsk_fastx *f) {
    if (f->finished) return 0;
    if (!f->started) {  // format check: first byte that is not '\n'
        for (;;) {
            if (f->r >= f->n && !fill(f)) {
                f->finished = true;
                return f->io_error ? -BSK_ERR_IO : 0;
            }
            const uint8_t c = f->buf[f->r++];
            if (c == '\n') continue;
            if (c == '>' || c == '@') {
                f->is_fastq = c == '@';
                f->delim = c;
                f->started = true;
                break;
            }
            f->finished = true;
            f->err = "fastx: invalid FASTA/Q format";
            return -BSK_ERR_NOT_FASTX;
        }
    }
    for (;;) {
        if (f->r >= f->n && !fill(f)) {  // end of file: what was collected is the last record
            f->finished = true;
            if (f->io_error) {  // not an end of file: the bytes collected so far are not a record
                f->rec.clear();
                return -BSK_ERR_IO;
            }
            const int st = parse(f);
            f->rec.clear();
            if (st == 3) return 0;
            if (st != 0) {
                f->err = "fastx: unequal sequence and quality";
                return -BSK_ERR_BAD_FASTQ;
            }
            return 1;
        }
        const uint8_t *b = f->buf.data() + f->r, *e = f->buf.data() + f->n;
        const uint8_t *d = (const uint8_t *)memchr(b, f->delim, (size_t)(e - b));
        if (!d) {
            f->rec.append((const char *)b, (size_t)(e - b));
            f->r = f->n;
            continue;
        }
        const uint8_t before = d > b ? d[-1] : (f->rec.empty() ? 0 : (uint8_t)f->rec.back());
        f->rec.append((const char *)b, (size_t)(d - b));
        f->r = (size_t)(d - f->buf.data()) + 1;
        if (before != '\n') {  // a delimiter character inside a line
            f->rec.push_back((char)f->delim);
            continue;
        }
        // the record ends before that "\n": the reference hands parseRecord the bytes without it (and without a '\r'
        // before it), reader.go:323 -- which matters for a malformed FASTQ record whose last line then counts as unterminated
        f->rec.pop_back();
        if (!f->rec.empty() && f->rec.back() == '\r') f->rec.pop_back();
        const int st = parse(f);
        if (st == 1) {  // '@' opened a quality line, not a record (reader.go:329-334)
            f->rec.push_back('\n');
            f->rec.push_back((char)f->delim);
            continue;
        }
        f->rec.clear();
        if (st == 2) {
            f->finished = true;
            f->err = "fastx: bad FASTQ format";
            return -BSK_ERR_BAD_FASTQ;
        }
        if (st == 3) {  // reader.go:440-442
            f->finished = true;
            return 0;
        }
        return 1;
    }
}

}  // namespace

extern "C" int bsk_fastx_open(const char *path, bsk_fastx **out) {
    if (!path || !out) return BSK_ERR_ARG;
    *out = nullptr;
    bsk_fastx *f = new (std::nothrow) bsk_fastx();
    if (!f) return BSK_ERR_NOMEM;
    f->fh = strcmp(path, "-") == 0 ? gzdopen(0, "rb") : gzopen(path, "rb");
    if (!f->fh) {
        delete f;
        return BSK_ERR_IO;
    }
    gzbuffer(f->fh, 1u << 20);
    const char *bs = getenv("BSK_FASTX_BUF");  // tests: tiny windows put every kind of boundary inside a record
    f->buf.resize(bs && atoi(bs) > 0 ? (size_t)atoi(bs) : (size_t)(4u << 20));
    *out = f;
    return BSK_OK;
}

extern "C" void bsk_fastx_close(bsk_fastx *f) {
    if (!f) return;
    if (f->fh) gzclose(f->fh);
    delete f;
}

extern "C" const char *bsk_fastx_error(const bsk_fastx *f) { return f ? f->err.c_str() : "null reader"; }

extern "C" int bsk_fastx_info(const bsk_fastx *f, int *is_fastq, int *alphabet) {
    if (!f) return BSK_ERR_ARG;
    if (is_fastq) *is_fastq = f->is_fastq;
    if (alphabet) *alphabet = f->alphabet == -2 ? -1 : f->alphabet;
    return BSK_OK;
}

extern "C" int bsk_fastx_read_chunk(bsk_fastx *f, uint64_t max_records, uint64_t max_bytes, uint64_t *n, const uint8_t **seq_bytes,
                                    const uint64_t **seq_offsets, const uint8_t **name_bytes, const uint64_t **name_offsets,
                                    const uint8_t **qual_bytes) {
    if (!f || !n) return BSK_ERR_ARG;
    f->seq.clear();
    f->name.clear();
    f->qual.clear();
    f->seq_off.assign(1, 0);
    f->name_off.assign(1, 0);
    uint64_t cnt = 0;
    int rc = f->pending;
    f->pending = 0;
    while (rc == BSK_OK && (max_records == 0 || cnt < max_records) && (max_bytes == 0 || f->seq.size() < max_bytes)) {
        const int st = next_record(f);
        if (st == 0) break;
        if (st < 0) {
            rc = -st;
            break;
        }
        if (f->alphabet == -2) f->alphabet = guess_alphabet(f->p_seq);
        f->seq.insert(f->seq.end(), f->p_seq.begin(), f->p_seq.end());
        f->name.insert(f->name.end(), f->p_head.begin(), f->p_head.end());
        if (f->is_fastq) f->qual.insert(f->qual.end(), f->p_qual.begin(), f->p_qual.end());
        f->seq_off.push_back(f->seq.size());
        f->name_off.push_back(f->name.size());
        ++cnt;
    }
    *n = cnt;
    if (f->seq.empty()) f->seq.push_back(0);  // never hand out a NULL data pointer
    if (f->name.empty()) f->name.push_back(0);
    if (seq_bytes) *seq_bytes = f->seq.data();
    if (seq_offsets) *seq_offsets = f->seq_off.data();
    if (name_bytes) *name_bytes = f->name.data();
    if (name_offsets) *name_offsets = f->name_off.data();
    if (f->is_fastq == 1 && f->qual.empty()) f->qual.push_back(0);
    if (qual_bytes) *qual_bytes = f->is_fastq == 1 ? f->qual.data() : nullptr;  // NULL = FASTA
    if (cnt && rc != BSK_OK) {  // hand out the good records now, the error with the next call
        f->pending = rc;
        rc = BSK_OK;
    }
    return rc;
}

extern "C" int bsk_batch_from_fastx(bsk_ctx *ctx, bsk_fastx *f, uint64_t max_records, uint64_t max_bytes, int alphabet, bsk_batch **out,
                                    uint64_t *n_records) {
    if (!ctx || !f || !out || !n_records) return BSK_ERR_ARG;
    *out = nullptr;
    const uint8_t *sb = nullptr;
    const uint64_t *so = nullptr;
    int rc = bsk_fastx_read_chunk(f, max_records, max_bytes, n_records, &sb, &so, nullptr, nullptr, nullptr);
    if (rc != BSK_OK || *n_records == 0) return rc;
    if (alphabet < 0) alphabet = f->alphabet;
    if (alphabet < BSK_ALPHA_DNA || alphabet > BSK_ALPHA_UNLIMIT) return BSK_ERR_UNSUPPORTED;  // guessed "Unlimit" (-1): the caller must say what it is
    return bsk_batch_from_ascii(ctx, sb, so, *n_records, alphabet, out);
}
