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
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "biosketch.h"

struct bsk_ctx;
struct bsk_batch;

// Byte ranges of a file for the block-parallel reader: a plain file (pread), or a BGZF file (the blocked gzip variant of htslib's
// bgzip: every gzip member holds at most 64 KiB and records its own compressed size in a "BC" extra field, so the members can be
// located without inflating anything and inflated independently) addressed by UNCOMPRESSED offset.
struct RangeFile {
    int fd = -1;
    uint64_t size = 0;                 // bytes of (uncompressed) text
    bool bgzf = false;
    std::vector<uint64_t> coff, uoff;  // BGZF: compressed / uncompressed offset of block i; one more entry = the end
};
struct RangeCursor {  // one per reading thread: the last inflated block
    z_stream zs;
    bool zs_ready = false;
    std::vector<uint8_t> comp, text;
    uint64_t block = ~0ULL;
    ~RangeCursor() {
        if (zs_ready) inflateEnd(&zs);
    }
};
// up to len bytes at (uncompressed) offset off; < 0: I/O or format error, 0: end of file
static ssize_t range_read(const RangeFile *rf, RangeCursor *cur, uint8_t *dst, size_t len, uint64_t off);
static bool range_read_full(const RangeFile *rf, RangeCursor *cur, uint8_t *dst, size_t len, uint64_t off) {
    while (len) {
        const ssize_t got = range_read(rf, cur, dst, len, off);
        if (got <= 0) return false;
        dst += got;
        len -= (size_t)got;
        off += (uint64_t)got;
    }
    return true;
}


struct bsk_fastx {
    gzFile fh = nullptr;
    const RangeFile *rf = nullptr;  // the other byte source (block-parallel reader below): range_read from `foff` on
    RangeCursor cursor;
    int fd = -1;
    uint64_t foff = 0;         // next file offset to read
    uint64_t win_off = 0;      // file offset of buf[0]
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
    if (f->rf) {  // a byte range of a plain or BGZF file
        const ssize_t got = range_read(f->rf, &f->cursor, f->buf.data(), f->buf.size(), f->foff);
        if (got <= 0) {
            if (got < 0) {
                f->io_error = true;
                f->err = std::string("fastx: read error: ") + (f->rf->bgzf && errno == 0 ? "damaged BGZF block" : strerror(errno));
            }
            f->eof = true;
            f->n = 0;
            return false;
        }
        f->win_off = f->foff;
        f->foff += (uint64_t)got;
        f->n = (size_t)got;
        return true;
    }
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

#ifndef BSK_FASTX_STANDALONE  // (the reader alone under the sanitizers: csrc/san-fastx -- the one entry that calls into the device library stays out)
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
#endif

// ------------------------------------------------------------------------------------------------------------------------
// Block-parallel reader for plain (uncompressed) files: the same records as the reader above, sequences only, parsed by
// n_threads threads that work ahead on consecutive byte ranges ("pieces") of the file.
//
// A piece is the records whose delimiter lies in [lo, hi).  Its thread guesses where the first of them starts -- FASTA: the
// first '>' after a newline, which always opens a record (reader.go:312-352); FASTQ: the first '@' after a newline whose
// next lines look like a record (third line '+', equal lengths, another '@' after the fourth), because a quality line may
// start with '@' too -- and then runs the ordinary record state machine (next_record above, over pread instead of gzread)
// from there until the first record that starts at or after `hi`.  The consumer hands the pieces out in file order and
// checks each guess against the previous piece's true end: a piece that started anywhere else (multi-line FASTQ defeats the
// guess, for instance) is parsed again from the right offset, serially.  So the result is that of the serial reader
// whatever the file looks like; only the speed depends on the guess.
// ------------------------------------------------------------------------------------------------------------------------
struct bsk_fastx_piece {
    uint64_t idx = 0;
    int64_t start = -1;      // file offset of the delimiter the parse started at; -1: no record start found in [lo, hi)
    uint64_t end = 0;        // file offset of the delimiter of the first record after the piece
    bool file_done = false;  // the reader finished inside this piece (end of file, an error, or reader.go:440-442)
    int err = BSK_OK;        // met after the records below
    std::string errtext;
    std::vector<uint8_t> seq;
    std::vector<uint64_t> off;
};

struct bsk_fastx_par {
    RangeFile rf;
    uint64_t fsize = 0, first = 0, piece_bytes = 0, n_pieces = 0;
    int is_fastq = -1, alphabet = -2;
    uint8_t delim = 0;
    size_t window = 0;
    std::string err;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    uint64_t next_idx = 0, consumed = 0, ahead = 0;
    std::map<uint64_t, bsk_fastx_piece *> done;
    std::vector<bsk_fastx_piece *> pool;
    bool stop = false;
    std::vector<std::thread> threads;
    uint64_t cur = 0;  // consumer: delimiter of the next record to deliver
    bool finished = false;
    int pending = 0;
    uint64_t reparsed = 0;  // pieces whose guessed start was wrong (statistics)
    bsk_fastx serial;       // the consumer's own record reader (re-parses)
};

namespace {

bool pread_full(int fd, uint8_t *dst, size_t len, uint64_t off) {
    while (len) {
        const ssize_t got = pread(fd, dst, len, (off_t)off);
        if (got < 0 && errno == EINTR) continue;
        if (got <= 0) return false;
        dst += got;
        len -= (size_t)got;
        off += (uint64_t)got;
    }
    return true;
}

// BGZF block header (SAM specification 4.1): gzip member with FEXTRA whose extra field holds the subfield 'B','C',2,BSIZE-1
bool bgzf_block_size(const uint8_t *h, size_t have, uint32_t *bsize, uint32_t *data_off) {
    if (have < 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return false;
    const uint32_t xlen = h[10] | (h[11] << 8);
    if (have < 12 + xlen) return false;
    for (uint32_t i = 0; i + 4 <= xlen;) {
        const uint8_t *sf = h + 12 + i;
        const uint32_t slen = sf[2] | (sf[3] << 8);
        if (sf[0] == 'B' && sf[1] == 'C' && slen == 2 && i + 6 <= xlen) {
            *bsize = (uint32_t)(sf[4] | (sf[5] << 8)) + 1;
            *data_off = 12 + xlen;
            return true;
        }
        i += 4 + slen;
    }
    return false;
}
// 1: a BGZF file, index built; 0: some other gzip file; < 0: I/O error
int bgzf_index(RangeFile *rf, uint64_t file_bytes) {
    uint64_t c = 0, u = 0;
    uint8_t h[512];
    while (c < file_bytes) {
        const size_t have = (size_t)std::min<uint64_t>(sizeof h, file_bytes - c);
        if (!pread_full(rf->fd, h, have, c)) return -1;
        uint32_t bsize = 0, doff = 0;
        if (!bgzf_block_size(h, have, &bsize, &doff) || bsize < doff + 8 || c + bsize > file_bytes) return rf->coff.empty() ? 0 : -1;
        uint8_t isz[4];
        if (!pread_full(rf->fd, isz, 4, c + bsize - 4)) return -1;
        const uint32_t isize = isz[0] | (isz[1] << 8) | (isz[2] << 16) | ((uint32_t)isz[3] << 24);
        if (isize > 65536) return -1;
        rf->coff.push_back(c);
        rf->uoff.push_back(u);
        c += bsize;
        u += isize;
    }
    rf->coff.push_back(c);
    rf->uoff.push_back(u);
    rf->size = u;
    rf->bgzf = true;
    return 1;
}

// does a FASTQ record plausibly start at b[k] ('@' after a newline)?  1 yes, 0 no, -1 the buffer ends too early to tell
int looks_like_fastq_record(const uint8_t *b, size_t len, size_t k, bool at_eof) {
    const uint8_t *e = b + len;
    const uint8_t *nl1 = (const uint8_t *)memchr(b + k, '\n', len - k);
    if (!nl1) return at_eof ? 1 : -1;
    const uint8_t *l2 = nl1 + 1;
    const uint8_t *nl2 = l2 < e ? (const uint8_t *)memchr(l2, '\n', (size_t)(e - l2)) : nullptr;
    if (!nl2) return at_eof ? 1 : -1;
    const uint8_t *l3 = nl2 + 1;
    if (l3 >= e) return at_eof ? 1 : -1;
    if (*l3 != '+') return 0;
    const uint8_t *nl3 = (const uint8_t *)memchr(l3, '\n', (size_t)(e - l3));
    if (!nl3) return at_eof ? 1 : -1;
    const uint8_t *l4 = nl3 + 1;
    const uint8_t *nl4 = l4 < e ? (const uint8_t *)memchr(l4, '\n', (size_t)(e - l4)) : nullptr;
    if (!nl4) return at_eof ? 1 : -1;
    size_t sl = (size_t)(nl2 - l2), ql = (size_t)(nl4 - l4);
    if (sl && l2[sl - 1] == '\r') --sl;
    if (ql && l4[ql - 1] == '\r') --ql;
    if (sl != ql) return 0;
    if (nl4 + 1 >= e) return at_eof ? 1 : -1;
    return nl4[1] == '@' ? 1 : 0;
}

// first guessed record start in [lo, hi) (lo > 0), or -1
int64_t find_start(bsk_fastx_par *p, RangeCursor *cur, uint64_t lo, uint64_t hi, std::vector<uint8_t> &tmp) {
    const uint64_t base = lo - 1;
    size_t want = 1u << 16;
    for (;;) {
        const size_t len = (size_t)std::min<uint64_t>(want, p->fsize - base);
        tmp.resize(len);
        if (!range_read_full(&p->rf, cur, tmp.data(), len, base)) return -1;  // the parse reports the I/O error
        const bool at_eof = base + len == p->fsize;
        const uint8_t *b = tmp.data();
        const size_t scan_end = (size_t)std::min<uint64_t>(len, hi - base);
        size_t i = 1;
        bool need_more = false;
        while (i < scan_end) {
            const uint8_t *d = (const uint8_t *)memchr(b + i, p->delim, scan_end - i);
            if (!d) break;
            const size_t k = (size_t)(d - b);
            if (b[k - 1] == '\n') {
                if (!p->is_fastq) return (int64_t)(base + k);
                const int ok = looks_like_fastq_record(b, len, k, at_eof);
                if (ok == 1) return (int64_t)(base + k);
                if (ok < 0) {
                    need_more = true;
                    break;
                }
            }
            i = k + 1;
        }
        if (!need_more && (base + len >= hi || at_eof)) return -1;
        if (at_eof) return -1;
        want *= 8;
    }
}

}  // namespace

static ssize_t range_read(const RangeFile *rf, RangeCursor *cur, uint8_t *dst, size_t len, uint64_t off) {
    errno = 0;
    if (off >= rf->size || len == 0) return 0;
    if (!rf->bgzf) {
        ssize_t got;
        do got = pread(rf->fd, dst, (size_t)std::min<uint64_t>(len, rf->size - off), (off_t)off);
        while (got < 0 && errno == EINTR);
        return got;
    }
    size_t done = 0;
    // block holding `off`: the last i with uoff[i] <= off (empty blocks share an offset: take the last of them that has data)
    size_t b = (size_t)(std::upper_bound(rf->uoff.begin(), rf->uoff.end(), off) - rf->uoff.begin()) - 1;
    while (done < len && off < rf->size) {
        while (b + 1 < rf->uoff.size() && rf->uoff[b + 1] <= off) ++b;
        if (b + 1 >= rf->uoff.size()) break;
        if (cur->block != b) {  // inflate block b
            const uint32_t bsize = (uint32_t)(rf->coff[b + 1] - rf->coff[b]);
            cur->comp.resize(bsize);
            if (!pread_full(rf->fd, cur->comp.data(), bsize, rf->coff[b])) return -1;
            uint32_t bs2 = 0, doff = 0;
            if (!bgzf_block_size(cur->comp.data(), bsize, &bs2, &doff) || bs2 != bsize) {
                errno = 0;
                return -1;
            }
            const uint32_t isize = (uint32_t)(rf->uoff[b + 1] - rf->uoff[b]);
            cur->text.resize(isize ? isize : 1);
            if (!cur->zs_ready) {
                memset(&cur->zs, 0, sizeof cur->zs);
                if (inflateInit2(&cur->zs, -15) != Z_OK) return -1;
                cur->zs_ready = true;
            } else {
                inflateReset(&cur->zs);
            }
            cur->zs.next_in = cur->comp.data() + doff;
            cur->zs.avail_in = bsize - doff - 8;
            cur->zs.next_out = cur->text.data();
            cur->zs.avail_out = isize;
            const int zr = inflate(&cur->zs, Z_FINISH);
            if (zr != Z_STREAM_END || cur->zs.avail_out != 0) {  // a damaged block
                cur->block = ~0ULL;
                errno = 0;
                return -1;
            }
            cur->block = b;
        }
        const uint64_t in_block = off - rf->uoff[b];
        const size_t take = (size_t)std::min<uint64_t>(len - done, rf->uoff[b + 1] - off);
        memcpy(dst + done, cur->text.data() + in_block, take);
        done += take;
        off += take;
    }
    return (ssize_t)done;
}

namespace {

void reader_for_range(bsk_fastx *f, const bsk_fastx_par *p, uint64_t from) {
    f->rf = &p->rf;
    f->foff = from + 1;  // the bytes after the delimiter
    f->win_off = 0;
    if (f->buf.size() != p->window) f->buf.resize(p->window);
    f->r = f->n = 0;
    f->eof = f->finished = f->io_error = false;
    f->started = true;
    f->is_fastq = p->is_fastq;
    f->delim = p->delim;
    f->pending = 0;
    f->rec.clear();
    f->err.clear();
}

// the records whose delimiter lies in [from, hi); `from` is the offset of a record's delimiter
void parse_span(bsk_fastx *f, const bsk_fastx_par *p, uint64_t from, uint64_t hi, bsk_fastx_piece *out) {
    reader_for_range(f, p, from);
    out->start = (int64_t)from;
    out->end = p->fsize;
    out->file_done = false;
    out->err = BSK_OK;
    out->errtext.clear();
    out->seq.clear();
    out->off.assign(1, 0);
    for (;;) {
        const int st = next_record(f);
        if (st <= 0) {
            out->file_done = true;
            if (st < 0) {
                out->err = -st;
                out->errtext = f->err;
            }
            return;
        }
        out->seq.insert(out->seq.end(), f->p_seq.begin(), f->p_seq.end());
        out->off.push_back(out->seq.size());
        if (f->finished) {  // that was the last record of the file
            out->file_done = true;
            return;
        }
        const uint64_t next_start = f->win_off + f->r - 1;  // next_record stopped right after the next record's delimiter
        if (next_start >= hi) {
            out->end = next_start;
            return;
        }
    }
}

void piece_range(const bsk_fastx_par *p, uint64_t idx, uint64_t &lo, uint64_t &hi) {
    lo = p->first + idx * p->piece_bytes;
    hi = std::min<uint64_t>(lo + p->piece_bytes, p->fsize);
}

void par_worker(bsk_fastx_par *p) {
    bsk_fastx rd;
    std::vector<uint8_t> tmp;
    for (;;) {
        bsk_fastx_piece *pc = nullptr;
        uint64_t idx;
        {
            std::unique_lock<std::mutex> l(p->m);
            p->cv_work.wait(l, [&] { return p->stop || (p->next_idx < p->n_pieces && p->next_idx < p->consumed + p->ahead); });
            if (p->stop) return;
            idx = p->next_idx++;
            if (!p->pool.empty()) {
                pc = p->pool.back();
                p->pool.pop_back();
            }
        }
        if (!pc) pc = new bsk_fastx_piece();
        pc->idx = idx;
        uint64_t lo, hi;
        piece_range(p, idx, lo, hi);
        try {
            const int64_t s = idx == 0 ? (int64_t)p->first : find_start(p, &rd.cursor, lo, hi, tmp);
            if (s >= 0) {
                parse_span(&rd, p, (uint64_t)s, hi, pc);
            } else {
                pc->start = -1;
                pc->seq.clear();
                pc->off.assign(1, 0);
                pc->err = BSK_OK;
                pc->file_done = false;
            }
        } catch (const std::bad_alloc &) {  // no exception leaves a thread (or crosses the C ABI): the consumer re-parses the piece
            pc->start = -1;                 // serially and meets the same condition in its own thread
            pc->seq.clear();
            pc->off.assign(1, 0);
            pc->err = BSK_OK;
            pc->file_done = false;
        }
        {
            std::lock_guard<std::mutex> l(p->m);
            p->done[idx] = pc;
        }
        p->cv_done.notify_all();
    }
}

}  // namespace

extern "C" int bsk_fastx_par_open(const char *path, int n_threads, uint64_t piece_bytes, bsk_fastx_par **out) {
    if (!path || !out || n_threads < 1 || n_threads > 256) return BSK_ERR_ARG;
    *out = nullptr;
    if (strcmp(path, "-") == 0) return BSK_ERR_UNSUPPORTED;  // a stream has no byte ranges
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return BSK_ERR_IO;
    struct stat sb;
    if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) {
        close(fd);
        return BSK_ERR_UNSUPPORTED;
    }
    bsk_fastx_par *p = new (std::nothrow) bsk_fastx_par();
    if (!p) {
        close(fd);
        return BSK_ERR_NOMEM;
    }
    p->rf.fd = fd;
    p->rf.size = (uint64_t)sb.st_size;
    uint8_t magic[2] = {0, 0};
    if (p->rf.size >= 2 && pread_full(fd, magic, 2, 0) && magic[0] == 0x1f && magic[1] == 0x8b) {
        // gzip: one serial stream -- unless it is BGZF, whose members are found by their recorded sizes and inflated independently
        const int bz = bgzf_index(&p->rf, (uint64_t)sb.st_size);
        if (bz != 1) {
            close(fd);
            delete p;
            return BSK_ERR_UNSUPPORTED;  // (a damaged BGZF file too: the serial reader reports where it breaks)
        }
    }
    p->fsize = p->rf.size;
    // format: the first byte that is not '\n' (reader.go:273-305)
    std::vector<uint8_t> tmp(1u << 16);
    RangeCursor cur0;
    uint64_t at = 0;
    bool found = false;
    while (at < p->fsize && !found) {
        const size_t len = (size_t)std::min<uint64_t>(tmp.size(), p->fsize - at);
        if (!range_read_full(&p->rf, &cur0, tmp.data(), len, at)) {
            close(fd);
            delete p;
            return BSK_ERR_IO;
        }
        for (size_t i = 0; i < len; ++i)
            if (tmp[i] != '\n') {
                if (tmp[i] != '>' && tmp[i] != '@') {
                    close(fd);
                    delete p;
                    return BSK_ERR_NOT_FASTX;
                }
                p->is_fastq = tmp[i] == '@';
                p->delim = tmp[i];
                p->first = at + i;
                found = true;
                break;
            }
        at += len;
    }
    const char *pb = getenv("BSK_FASTX_PIECE");  // tests: tiny pieces put a piece boundary at every kind of place
    p->piece_bytes = piece_bytes ? piece_bytes : (pb && atoll(pb) > 0 ? (uint64_t)atoll(pb) : (uint64_t)(8u << 20));
    const char *bs = getenv("BSK_FASTX_BUF");
    p->window = bs && atoi(bs) > 0 ? (size_t)atoi(bs) : (size_t)(1u << 20);
    p->n_pieces = found ? (p->fsize - p->first + p->piece_bytes - 1) / p->piece_bytes : 0;  // no record at all: nothing to do
    p->cur = p->first;
    p->ahead = 2 * (uint64_t)n_threads + 2;
    p->finished = !found;
    for (int t = 0; t < n_threads && (uint64_t)t < p->n_pieces; ++t) p->threads.emplace_back(par_worker, p);
    *out = p;
    return BSK_OK;
}

extern "C" void bsk_fastx_par_close(bsk_fastx_par *p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> l(p->m);
        p->stop = true;
    }
    p->cv_work.notify_all();
    for (auto &t : p->threads) t.join();
    for (auto &kv : p->done) delete kv.second;
    for (auto *pc : p->pool) delete pc;
    if (p->rf.fd >= 0) close(p->rf.fd);
    delete p;
}

extern "C" const char *bsk_fastx_par_error(const bsk_fastx_par *p) { return p ? p->err.c_str() : "null reader"; }

extern "C" int bsk_fastx_par_info(const bsk_fastx_par *p, int *is_fastq, int *alphabet, uint64_t *reparsed_pieces) {
    if (!p) return BSK_ERR_ARG;
    if (is_fastq) *is_fastq = p->is_fastq;
    if (alphabet) *alphabet = p->alphabet == -2 ? -1 : p->alphabet;
    if (reparsed_pieces) *reparsed_pieces = p->reparsed;
    return BSK_OK;
}

extern "C" void bsk_fastx_piece_release(bsk_fastx_par *p, bsk_fastx_piece *pc) {
    if (!p || !pc) return;
    std::lock_guard<std::mutex> l(p->m);
    p->pool.push_back(pc);
}

extern "C" int bsk_fastx_piece_data(const bsk_fastx_piece *pc, uint64_t *n, const uint8_t **seq_bytes, const uint64_t **seq_offsets) {
    if (!pc || !n) return BSK_ERR_ARG;
    *n = pc->off.size() - 1;
    static const uint8_t zero = 0;
    if (seq_bytes) *seq_bytes = pc->seq.empty() ? &zero : pc->seq.data();
    if (seq_offsets) *seq_offsets = pc->off.data();
    return BSK_OK;
}

extern "C" int bsk_fastx_par_next(bsk_fastx_par *p, bsk_fastx_piece **piece) {
    if (!p || !piece) return BSK_ERR_ARG;
    *piece = nullptr;
    if (p->pending) {
        const int rc = p->pending;
        p->pending = 0;
        p->finished = true;
        return rc;
    }
    for (;;) {
        if (p->finished || p->consumed >= p->n_pieces) return BSK_OK;
        bsk_fastx_piece *pc;
        {
            std::unique_lock<std::mutex> l(p->m);
            p->cv_done.wait(l, [&] { return p->done.count(p->consumed) != 0; });
            auto it = p->done.find(p->consumed);
            pc = it->second;
            p->done.erase(it);
            p->consumed++;
        }
        p->cv_work.notify_all();
        uint64_t lo, hi;
        piece_range(p, pc->idx, lo, hi);
        if (p->cur >= hi) {  // the previous piece's last record runs over this whole range
            bsk_fastx_piece_release(p, pc);
            continue;
        }
        if (pc->start != (int64_t)p->cur) {  // no guess or a wrong one: the records of [cur, hi), serially
            p->reparsed++;
            try {
                parse_span(&p->serial, p, p->cur, hi, pc);
            } catch (const std::bad_alloc &) {
                p->err = "fastx: out of memory";
                p->finished = true;
                bsk_fastx_piece_release(p, pc);
                return BSK_ERR_NOMEM;
            }
        }
        if (pc->file_done) p->finished = true;
        else p->cur = pc->end;
        const uint64_t n = pc->off.size() - 1;
        if (p->alphabet == -2 && n) p->alphabet = guess_alphabet(std::string((const char *)pc->seq.data(), (size_t)pc->off[1]));
        if (pc->err != BSK_OK) {
            p->err = pc->errtext;
            if (n == 0) {
                const int rc = pc->err;
                p->finished = true;
                bsk_fastx_piece_release(p, pc);
                return rc;
            }
            p->pending = pc->err;  // the good records now, the error with the next call (as bsk_fastx_read_chunk)
        }
        if (n == 0) {
            bsk_fastx_piece_release(p, pc);
            continue;
        }
        *piece = pc;
        return BSK_OK;
    }
}
