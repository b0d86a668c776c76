/*
 * bio_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of shenwei356/bio `sketches/` (reference v0.13.8).
 * Every function cites the reference file:line it follows.  See bio_oracle.h
 * for the parity/pinning status of each part.  Nothing under bio_amd/ may
 * include, link or call this file.
 */
#include "bio_oracle.h"

#include <stdlib.h>
#include <string.h>

/* =====================================================================
 * ntHash-1.  The arithmetic lives in github.com/will-rowe/nthash v0.4.0
 * (go.mod:14), which is NOT vendored.  Restated from the published ntHash
 * (Mohamadi et al. 2016, nthash.hpp v1) that the Go module ports; pinned
 * for A/C/G/T by sketches/sketch_test.go:67-72 (tests/test_oracle_golden.py).
 * Call sites: iterator.go:159,279,442,649,659; sketch.go:120,179,184,212,319,344,367.
 * ===================================================================== */
#define SEED_A 0x3c8bfbb395c60474ULL
#define SEED_C 0x3193c18562a02b4cULL
#define SEED_G 0x20323ed082572324ULL
#define SEED_T 0x295549f54be24456ULL
#define SEED_N 0x0000000000000000ULL

/* forward seed of a byte: A/a C/c G/g T/t U/u, everything else 0 */
uint64_t orc_seed_fwd(uint8_t b) {
    switch (b) {
    case 'A': case 'a': return SEED_A;
    case 'C': case 'c': return SEED_C;
    case 'G': case 'g': return SEED_G;
    case 'T': case 't': case 'U': case 'u': return SEED_T;
    default: return SEED_N;
    }
}

/* ntHash-1 looks the complement up as seedTab[b & 0x07]; entries 0..7 of the
 * published table are {N, T, N, G, A, A, N, C}. ('A'&7=1 -> T, 'C'&7=3 -> G,
 * 'G'&7=7 -> C, 'T'&7=4 -> A, 'U'&7=5 -> A, 'N'&7=6 -> 0.) */
uint64_t orc_seed_rev(uint8_t b) {
    static const uint64_t low8[8] = {SEED_N, SEED_T, SEED_N, SEED_G, SEED_A, SEED_A, SEED_N, SEED_C};
    return low8[b & 0x07];
}

static inline uint64_t rol64(uint64_t v, unsigned n) {
    n &= 63u;
    return n ? (v << n) | (v >> (64u - n)) : v;
}
static inline uint64_t ror64(uint64_t v, unsigned n) {
    n &= 63u;
    return n ? (v >> n) | (v << (64u - n)) : v;
}

static inline int is_acgt(uint8_t b) {
    switch (b) {
    case 'A': case 'a': case 'C': case 'c': case 'G': case 'g': case 'T': case 't': return 1;
    default: return 0;
    }
}

/* nthash.NewHasher: hashes k-mer 0 on both strands
 * (fh = XOR_j rol(seed[s_j], k-1-j); rh = XOR_j rol(seed[comp s_j], j)). */
int orc_nthi_init(orc_nthi *h, const uint8_t *seq, size_t len, unsigned k) {
    if (k == 0 || k > len) return ORC_ERR_SHORT_SEQ;
    h->seq = seq;
    h->len = len;
    h->k = k;
    uint64_t fh = 0, rh = 0;
    for (unsigned i = 0; i < k; i++) {
        fh = rol64(fh, 1) ^ orc_seed_fwd(seq[i]);
        rh = rol64(rh, 1) ^ orc_seed_rev(seq[k - 1 - i]);
    }
    h->fh = fh;
    h->rh = rh;
    h->cur = 0;
    h->max_idx = len - (k - 1);
    return ORC_OK;
}

/* (*NTHi).Next(canonical): roll by one base, return min(fh, rh) or fh. */
int orc_nthi_next(orc_nthi *h, int canonical, uint64_t *hash, int *strand) {
    if (h->cur >= h->max_idx) return 0;
    if (h->cur != 0) {
        uint8_t prev = h->seq[h->cur - 1];
        uint8_t end = h->seq[h->cur + h->k - 1];
        h->fh = rol64(h->fh, 1) ^ rol64(orc_seed_fwd(prev), h->k) ^ orc_seed_fwd(end);
        h->rh = ror64(h->rh, 1) ^ ror64(orc_seed_rev(prev), 1) ^ rol64(orc_seed_rev(end), h->k - 1);
    }
    h->cur++;
    if (canonical && h->rh < h->fh) {
        *hash = h->rh;
        if (strand) *strand = 1;
    } else {
        *hash = h->fh;
        if (strand) *strand = 0;
    }
    return 1;
}

/* circular: the reference appends the first k-1 bases to a copy
 * (iterator.go:642-646, sketch.go:106-110,163-167). Returns malloc'd buffer. */
static uint8_t *dup_seq(const uint8_t *seq, size_t len, int k, int circular, size_t *newlen) {
    size_t extra = circular ? (size_t)(k - 1) : 0;
    uint8_t *p = (uint8_t *)malloc(len + extra + 1);
    if (!p) return NULL;
    memcpy(p, seq, len);
    if (extra) memcpy(p + len, seq, extra);
    *newlen = len + extra;
    return p;
}

/* ---- A2: NewHashIterator iterator.go:615-655, NextHash :658-665 ---- */
long long orc_nthash_all(const uint8_t *seq, size_t len, int k, int canonical, int circular,
                         uint64_t *out, uint8_t *strand, size_t cap) {
    if (k < 1) return ORC_ERR_INVALID_K;               /* iterator.go:616 */
    if (len < (size_t)k) return ORC_ERR_SHORT_SEQ;     /* iterator.go:619 */
    size_t L;
    uint8_t *s = dup_seq(seq, len, k, circular, &L);
    if (!s) return ORC_ERR_NOMEM;
    orc_nthi h;
    orc_nthi_init(&h, s, L, (unsigned)k);
    long long n = 0;
    uint64_t code;
    int st;
    while (orc_nthi_next(&h, canonical, &code, &st)) {
        if (out || strand) {
            if ((size_t)n >= cap) { free(s); return ORC_ERR_CAPACITY; }
            if (out) out[n] = code;
            if (strand) strand[n] = (uint8_t)st;
        }
        n++;
    }
    free(s);
    return n;
}

/* ---- A1: k-mer codes.  base2bit: sketches/kmers.go:23-40 (A0 C1 G2 T3,
 * IUPAC -> one member, everything else 4). ---- */
static const uint8_t base2bit[256] = {
    4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
    4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
    4, 0, 1, 1, 0, 4, 4, 2, 0, 4, 4, 2, 4, 0, 0, 4, 4, 4, 0, 1, 3, 3, 0, 0, 4, 1, 4, 4, 4, 4, 4, 4,
    4, 0, 1, 1, 0, 4, 4, 2, 0, 4, 4, 2, 4, 0, 0, 4, 4, 4, 0, 1, 3, 3, 0, 0, 4, 1, 4, 4, 4, 4, 4, 4,
    4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
    4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
    4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
    4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4};

/* seq.RevComInplace (seq/seq.go:350) with the pair table of the sequence's own alphabet
 * (seq/alphabet.go:353-383; `alphabet` numbered as in include/biosketch.h: 0 DNAredundant,
 * 2 DNA, 3 RNA, 4 RNAredundant, 5 Unlimit = ComplementInplace returns at once,
 * seq/seq.go:381-383); gap / ambiguous letters pair with themselves (alphabet.go:155-160) and bytes
 * without a pair stay unchanged (PairLetter's error is ignored, seq/seq.go:390). */
static uint8_t dna_pair(uint8_t b, int alphabet) {
    static const char *const from[] = {"acgtryswkmbdhvACGTRYSWKMBDHV", "", "acgtACGT", "acguACGU", "acguryswkmbdhvACGURYSWKMBDHV", ""};
    static const char *const to[] = {"tgcayrswmkvhdbTGCAYRSWMKVHDB", "", "tgcaTGCA", "ugcaUGCA", "ugcayrswmkvhdbUGCAYRSWMKVHDB", ""};
    if (alphabet < 0 || alphabet > 5) alphabet = 0;
    for (int i = 0; from[alphabet][i]; i++)
        if ((uint8_t)from[alphabet][i] == b) return (uint8_t)to[alphabet][i];
    return b;
}
static void revcom_inplace(uint8_t *s, size_t n, int alphabet) {
    for (size_t i = 0, j = n; i + 1 < j; i++) {
        j--;
        uint8_t t = s[i];
        s[i] = s[j];
        s[j] = t;
    }
    for (size_t i = 0; i < n; i++) s[i] = dna_pair(s[i], alphabet);
}

/* NextKmer iterator.go:708-759.  first k-mer: kmers.Encode + kmers.MustRevComp
 * (shenwei356/kmers v0.1.0, un-vendored; encoding fixed by the rolling formulas
 * iterator.go:736,740: first base in the most significant pair). */
long long orc_kmer_all(const uint8_t *seq, size_t len, int k, int canonical, int circular,
                       uint64_t *out, size_t cap) {
    return orc_kmer_all_alpha(seq, len, k, canonical, circular, 0, out, cap);
}
long long orc_kmer_all_alpha(const uint8_t *seq, size_t len, int k, int canonical, int circular, int alphabet,
                             uint64_t *out, size_t cap) {
    if (k < 1) return ORC_ERR_INVALID_K;            /* iterator.go:669 */
    if (len < (size_t)k) return ORC_ERR_SHORT_SEQ;  /* iterator.go:672 */
    if (k > 32) return ORC_ERR_K_TOO_LARGE;         /* kmers.Encode -> ErrKOverflow at first Next */
    size_t L;
    uint8_t *s = dup_seq(seq, len, k, circular, &L);
    if (!s) return ORC_ERR_NOMEM;
    const unsigned kp1 = (unsigned)(k - 1);
    const uint64_t mask1 = (kp1 * 2 >= 64) ? ~0ULL : ((1ULL << (kp1 * 2)) - 1); /* iterator.go:698 */
    const unsigned mask2 = kp1 * 2;                                             /* iterator.go:699 */
    long long n = 0;
    size_t end = L - (size_t)k + 1;
    for (int strand = 0; strand < (canonical ? 1 : 2); strand++) {
        if (strand == 1) revcom_inplace(s, L, alphabet); /* iterator.go:719 */
        uint64_t pre = 0, preRC = 0;
        for (size_t idx = 0; idx < end; idx++) {
            uint64_t code, rc;
            if (idx != 0) {
                uint64_t b = base2bit[s[idx + kp1]];
                if (b == 4) { free(s); return ORC_ERR_ILLEGAL_BASE; } /* iterator.go:731-733,746 */
                code = ((pre & mask1) << 2) | b;                       /* iterator.go:736 */
                rc = ((b ^ 3) << mask2) | (preRC >> 2);                /* iterator.go:740 */
            } else {
                code = 0;
                rc = 0;
                for (int j = 0; j < k; j++) {
                    uint64_t b = base2bit[s[j]];
                    if (b == 4) { free(s); return ORC_ERR_ILLEGAL_BASE; }
                    code = (code << 2) | b;
                    rc |= (b ^ 3) << (2 * j);
                }
            }
            pre = code;
            preRC = rc;
            if (canonical && code > rc) code = rc; /* iterator.go:754 */
            if (out) {
                if ((size_t)n >= cap) { free(s); return ORC_ERR_CAPACITY; }
                out[n] = code;
            }
            n++;
        }
    }
    free(s);
    return n;
}

/* ---- A3: SimHash iterator.go:113-612 ---- */
long long orc_simhash_all(const uint8_t *seq, size_t len, int k, int m, int scale, int canonical,
                          int circular, uint64_t *out, size_t cap) {
    if (k < 1) return ORC_ERR_INVALID_K;                           /* :114 */
    if (k >= 65535) return ORC_ERR_K_TOO_LARGE;                    /* :117 */
    if (m < 4 || m > k) return ORC_ERR_INVALID_M;                  /* :121 */
    if (scale < 1 || scale > k - m + 1) return ORC_ERR_INVALID_SCALE; /* :124 */
    if (len < (size_t)k) return ORC_ERR_SHORT_SEQ;                 /* :128 */
    size_t L;
    uint8_t *s = dup_seq(seq, len, k, circular, &L);
    if (!s) return ORC_ERR_NOMEM;
    const int nh = k - m + 1;
    uint64_t *hashes = (uint64_t *)calloc((size_t)nh, sizeof(uint64_t));
    if (!hashes) { free(s); return ORC_ERR_NOMEM; }
    orc_nthi h;
    orc_nthi_init(&h, s, L, (unsigned)m); /* :159 hasher over m-mers */
    int16_t sum[64];
    memset(sum, 0, sizeof sum);
    int16_t npos = 0;
    const int frac = scale > 1;                                          /* :180 */
    const uint64_t maxhash = frac ? UINT64_MAX / (uint64_t)scale : UINT64_MAX; /* :181-185 */
    int pre_i = 0;
    long long n = 0;
    size_t end = L - (size_t)k + 1;
    for (size_t idx = 0; idx < end; idx++) {
        uint64_t hv = 0;
        if (idx != 0) {
            uint64_t pre = hashes[pre_i]; /* :204 */
            if (pre > 0) {
                npos--;
                for (int b = 0; b < 64; b++) sum[b] = (int16_t)(sum[b] - (int16_t)((pre >> (63 - b)) & 1));
            }
            hv = 0;
            orc_nthi_next(&h, canonical, &hv, NULL); /* :279 (ok ignored) */
            if (frac && hv > maxhash) hv = 0;        /* :281 */
            else if (hv > 0) npos++;
            hashes[pre_i] = hv;                      /* :288 */
            if (hv > 0)
                for (int b = 0; b < 64; b++) sum[b] = (int16_t)(sum[b] + (int16_t)((hv >> (63 - b)) & 1));
            pre_i = (pre_i == k - m) ? 0 : pre_i + 1; /* :434-438 */
        } else {
            npos = 0;
            for (int j = 0; j <= k - m; j++) { /* :441 */
                hv = 0;
                orc_nthi_next(&h, canonical, &hv, NULL);
                if (frac && hv > maxhash) { hashes[j] = 0; continue; }
                hashes[j] = hv;
                if (hv == 0) continue;
                npos++;
                for (int b = 0; b < 64; b++) sum[b] = (int16_t)(sum[b] + (int16_t)((hv >> (63 - b)) & 1));
            }
            pre_i = 0;
        }
        uint64_t code = 0;
        int16_t thr = (int16_t)((npos + 1) / 2); /* :360 */
        if (npos > 0) {
            for (int b = 0; b < 64; b++) {
                int16_t d = (int16_t)(sum[b] - thr);
                uint64_t bit = (uint64_t)((((uint16_t)d) >> 15) & 1u) ^ 1u; /* :365 sign-bit trick */
                code |= bit << (63 - b);
            }
        }
        if (out) {
            if ((size_t)n >= cap) { free(hashes); free(s); return ORC_ERR_CAPACITY; }
            out[n] = code;
        }
        n++;
    }
    free(hashes);
    free(s);
    return n;
}

/* =====================================================================
 * Sketch state machine (sketch.go:45-77).
 * ===================================================================== */
struct orc_sketch {
    uint8_t *S;
    size_t len;
    int borrowed; /* S points into the caller's buffer */
    int bufcap;   /* entries allocated for buf */
    int k, s, w, r, kMs;
    int minimizer, skip;
    orc_nthi hasher, hasherS;
    long long idx, end;
    long long mI;
    uint64_t mV;
    long long preMinIdx;
    orc_idxval *buf;
    int buflen;
    long long *pre; /* preMinIdxs */
    int prelen, precap;
    long long bsyncmerIdx;
    int late;
    unsigned flags;
    int done;
};

/* stands in for sorts.Quicksort(idxValues(buf)) (sketch.go:236,351): orders by
 * Val only.  Upstream is unstable (PARITY UNPINNED on ties); here: stable
 * insertion sort + ORC_FLAG_FIRST_WINDOW_TIE (same rule as tie_flag() below: a tied pair
 * with nothing smaller behind its first entry inside this window). */
static void first_window_sort(orc_idxval *buf, int n, unsigned *flags) {
    for (int a = 0; a < n; a++) /* buf is still in Idx order here */
        for (int b = a + 1; b < n; b++) {
            if (buf[a].val != buf[b].val) continue;
            int smaller = 0;
            for (int c = a + 1; c < n; c++)
                if (buf[c].val < buf[a].val) { smaller = 1; break; }
            if (!smaller) *flags |= ORC_FLAG_FIRST_WINDOW_TIE;
        }
    for (int i = 1; i < n; i++) {
        orc_idxval x = buf[i];
        int j = i - 1;
        while (j >= 0 && buf[j].val > x.val) {
            buf[j + 1] = buf[j];
            j--;
        }
        buf[j + 1] = x;
    }
}

static void scan_non_acgt(const uint8_t *s, size_t n, unsigned *flags) {
    for (size_t i = 0; i < n; i++)
        if (!is_acgt(s[i])) { *flags |= ORC_FLAG_HAS_NON_ACGT; return; }
}

/* the evict step shared by both sketches: sketch.go:250-258 / :355-363 */
static void buf_evict(orc_sketch *s, long long victim) {
    for (int i = 0; i < s->buflen; i++) {
        if (s->buf[i].idx == victim) {
            if (i < s->r) memmove(&s->buf[i], &s->buf[i + 1], (size_t)(s->r - i) * sizeof(orc_idxval));
            s->buflen = s->r;
            break;
        }
    }
}

/* the insert step: binary search + shift, sketch.go:261-295 / :371-405 */
static void buf_insert(orc_sketch *s, long long idx, uint64_t code) {
    int flag = 0, i = 0;
    int b = 0, e = s->r - 1, t;
    orc_idxval *buf = s->buf;
    for (;;) {
        t = b + (e - b) / 2;
        if (code < buf[t].val) {
            e = t - 1;
            if (e <= b) { flag = 1; i = b; break; }
        } else {
            b = t + 1;
            if (b >= s->r) { flag = 0; break; }
            if (b >= e) { flag = 1; i = e; break; }
        }
    }
    if (!flag) {
        buf[s->buflen].idx = idx;
        buf[s->buflen].val = code;
        s->buflen++;
    } else {
        if (code >= buf[i].val) i++;
        memmove(&buf[i + 1], &buf[i], (size_t)(s->r - i) * sizeof(orc_idxval));
        s->buflen++;
        buf[i].idx = idx;
        buf[i].val = code;
    }
}

/* reuse != NULL: a sketch that has finished, taken back "from the pool" (poolSketch, sketch.go:79: the Go objects are
 * recycled, their buffers with them); borrow: keep a pointer to the caller's bytes as the reference does (`S.Seq`, copied
 * only when circular, sketch.go:106-110) instead of the private copy the test-facing constructors make. */
static orc_sketch *sketch_alloc(orc_sketch *reuse, int borrow, const uint8_t *seq, size_t len, int k, int circular, int bufcap) {
    orc_sketch *s = reuse;
    if (s) {
        orc_idxval *buf = s->buf;
        long long *pre = s->pre;
        const int bc = s->bufcap, pc = s->precap;
        if (!s->borrowed) free(s->S);
        memset(s, 0, sizeof *s);
        s->buf = buf;
        s->bufcap = bc;
        s->pre = pre;
        s->precap = pc;
    } else {
        s = (orc_sketch *)calloc(1, sizeof *s);
        if (!s) return NULL;
    }
    if (borrow && !circular) {
        s->S = (uint8_t *)(uintptr_t)seq;
        s->len = len;
        s->borrowed = 1;
    } else {
        s->S = dup_seq(seq, len, k, circular, &s->len);
    }
    if (s->bufcap < bufcap + 2) {
        free(s->buf);
        s->buf = (orc_idxval *)calloc((size_t)bufcap + 2, sizeof(orc_idxval));
        s->bufcap = bufcap + 2;
    }
    if (!s->pre) {
        s->precap = 8;
        s->pre = (long long *)calloc((size_t)s->precap, sizeof(long long));
    }
    if (!s->S || !s->buf || !s->pre) { orc_sketch_free(s); return NULL; }
    return s;
}

void orc_sketch_free(orc_sketch *s) {
    if (!s) return;
    if (!s->borrowed) free(s->S);
    free(s->buf);
    free(s->pre);
    free(s);
}

/* NewMinimizerSketch sketch.go:85-138 */
static int minimizer_init(orc_sketch *reuse, int borrow, const uint8_t *seq, size_t len, int k, int w, int circular, orc_sketch **out) {
    *out = NULL;
    if (k < 1) return ORC_ERR_INVALID_K;                                  /* :86 */
    if (w < 1) return ORC_ERR_INVALID_W;                                  /* :89 (w <= 2^31-1 by type) */
    if (len < (size_t)k + (size_t)w - 1) return ORC_ERR_SHORT_SEQ;        /* :92 */
    orc_sketch *s = sketch_alloc(reuse, borrow, seq, len, k, circular, w);
    if (!s) return ORC_ERR_NOMEM;
    s->minimizer = 1;
    s->k = k;
    s->w = w;
    s->skip = (w == 1);
    s->idx = 0;
    s->end = (long long)s->len - 1; /* :114 */
    s->r = w - 1;                   /* :115 */
    orc_nthi_init(&s->hasher, s->S, s->len, (unsigned)k);
    s->preMinIdx = -1;
    if (!borrow) scan_non_acgt(s->S, s->len, &s->flags); /* oracle-only flag; the reference makes no such pass */
    *out = s;
    return ORC_OK;
}
int orc_minimizer_new(const uint8_t *seq, size_t len, int k, int w, int circular, orc_sketch **out) {
    return minimizer_init(NULL, 0, seq, len, k, w, circular, out);
}

/* NewSyncmerSketch sketch.go:142-202 */
static int syncmer_init(orc_sketch *reuse, int borrow, const uint8_t *seq, size_t len, int k, int sm, int circular, orc_sketch **out) {
    *out = NULL;
    if (k < 1) return ORC_ERR_INVALID_K;                      /* :143 */
    if (sm > k || sm <= 0) return ORC_ERR_INVALID_S;          /* :146 (s==0; negative s would panic upstream) */
    if ((long long)len < 2LL * k - sm - 1) return ORC_ERR_SHORT_SEQ; /* :149 */
    orc_sketch *s = sketch_alloc(reuse, borrow, seq, len, k, circular, (k - sm) * 2);
    if (!s) return ORC_ERR_NOMEM;
    s->minimizer = 0;
    s->k = k;
    s->s = sm;
    s->skip = (sm == k);
    s->idx = 0;
    s->end = (long long)s->len - 2LL * k + sm + 1; /* :170 */
    s->r = 2 * k - sm - 1 - sm;                    /* :171 */
    s->kMs = k - sm;
    s->w = k - sm;
    if (orc_nthi_init(&s->hasher, s->S, s->len, (unsigned)k) != ORC_OK ||
        orc_nthi_init(&s->hasherS, s->S, s->len, (unsigned)sm) != ORC_OK) {
        if (!reuse) orc_sketch_free(s); /* a recycled object stays with its owner */
        return ORC_ERR_SHORT_SEQ;
    }
    s->preMinIdx = -1;
    if (!borrow) scan_non_acgt(s->S, s->len, &s->flags);
    *out = s;
    return ORC_OK;
}
int orc_syncmer_new(const uint8_t *seq, size_t len, int k, int sm, int circular, orc_sketch **out) {
    return syncmer_init(NULL, 0, seq, len, k, sm, circular, out);
}

/* NextMinimizer sketch.go:205-309 */
static int next_minimizer(orc_sketch *s, uint64_t *out) {
    uint64_t code;
    for (;;) {
        if (s->idx > s->end) return 0;                                /* :207 */
        if (!orc_nthi_next(&s->hasher, 1, &code, NULL)) return 0;     /* :212 */
        if (s->skip) {                                                /* :218 */
            s->mI = s->idx;
            s->idx++;
            *out = code;
            return 1;
        }
        if (s->idx < s->r) { /* :225 */
            s->buf[s->buflen].idx = s->idx;
            s->buf[s->buflen].val = code;
            s->buflen++;
            s->idx++;
            continue;
        }
        if (s->idx == s->r) { /* :233 */
            s->buf[s->buflen].idx = s->idx;
            s->buf[s->buflen].val = code;
            s->buflen++;
            first_window_sort(s->buf, s->buflen, &s->flags);
            s->mI = s->buf[0].idx;
            s->mV = s->buf[0].val;
            s->preMinIdx = s->mI;
            s->idx++;
            *out = s->mV;
            return 1;
        }
        buf_evict(s, s->idx - s->w);    /* :250 */
        buf_insert(s, s->idx, code);    /* :261 */
        if (s->buf[0].idx == s->preMinIdx) { /* :298 */
            s->idx++;
            continue;
        }
        s->mI = s->buf[0].idx;
        s->mV = s->buf[0].val;
        s->preMinIdx = s->mI;
        s->idx++;
        *out = s->mV;
        return 1;
    }
}

static void pre_pop_front(orc_sketch *s) {
    memmove(&s->pre[0], &s->pre[1], (size_t)(s->prelen - 1) * sizeof(long long));
    s->prelen--;
}
static int pre_push(orc_sketch *s, long long v) {
    if (s->prelen == s->precap) {
        long long *p = (long long *)realloc(s->pre, (size_t)s->precap * 2 * sizeof(long long));
        if (!p) return 0;
        s->pre = p;
        s->precap *= 2;
    }
    s->pre[s->prelen++] = v;
    return 1;
}

/* NextSyncmer sketch.go:312-477 */
static int next_syncmer(orc_sketch *s, uint64_t *out) {
    uint64_t code, v;
    for (;;) {
        if (s->idx > s->end) return 0;                            /* :314 */
        if (!orc_nthi_next(&s->hasher, 1, &code, NULL)) return 0; /* :319 */
        if (s->skip) {                                            /* :328 */
            s->idx++;
            *out = code;
            return 1;
        }
        s->late = (s->prelen > 0 && s->idx == s->pre[0]); /* :333 */

        if (s->idx == 0) { /* :341 */
            for (long long i = s->idx; i <= s->idx + s->r; i++) {
                if (!orc_nthi_next(&s->hasherS, 1, &v, NULL)) return 0;
                s->buf[s->buflen].idx = i;
                s->buf[s->buflen].val = v;
                s->buflen++;
            }
            first_window_sort(s->buf, s->buflen, &s->flags); /* :351 */
        } else {
            buf_evict(s, s->idx - 1);                                   /* :355 */
            if (!orc_nthi_next(&s->hasherS, 1, &v, NULL)) return 0;     /* :367 */
            buf_insert(s, s->idx + s->r, v);                            /* :371 */
        }
        s->mI = s->buf[0].idx; /* :408 */
        s->mV = s->buf[0].val;

        if (s->mI - s->idx < s->w) s->bsyncmerIdx = s->mI; /* :413 */
        else s->bsyncmerIdx = s->mI - s->kMs;

        if (s->prelen > 0 && s->bsyncmerIdx == s->pre[0]) { /* :424 duplicated */
            if (s->late) {
                pre_pop_front(s);
                s->idx++;
                s->preMinIdx = s->bsyncmerIdx;
                *out = code;
                return 1;
            }
            s->idx++;
            continue;
        }
        if (s->late) { /* :441 */
            pre_pop_front(s);
            if (s->preMinIdx != s->bsyncmerIdx) pre_push(s, s->bsyncmerIdx);
            s->idx++;
            s->preMinIdx = s->bsyncmerIdx;
            *out = code;
            return 1;
        }
        if (s->bsyncmerIdx == s->idx) { /* :457 */
            if (s->prelen > 0) pre_pop_front(s);
            s->idx++;
            s->preMinIdx = s->bsyncmerIdx;
            *out = code;
            return 1;
        }
        if (s->preMinIdx != s->bsyncmerIdx) pre_push(s, s->bsyncmerIdx); /* :470 */
        s->idx++;
        s->preMinIdx = s->bsyncmerIdx;
    }
}

int orc_sketch_next(orc_sketch *s, uint64_t *code) { /* Next sketch.go:480 */
    if (s->done) return 0;
    int ok = s->minimizer ? next_minimizer(s, code) : next_syncmer(s, code);
    if (!ok) s->done = 1;
    return ok;
}
long long orc_sketch_index(const orc_sketch *s) { /* Index sketch.go:488 */
    return s->minimizer ? s->mI : s->idx - 1;
}
unsigned orc_sketch_flags(const orc_sketch *s) { return s->flags; }

static long long drain(orc_sketch *sk, int k, uint64_t *hash, uint32_t *pos, uint8_t *strand,
                       size_t cap, unsigned *flags) {
    long long n = 0;
    uint64_t code;
    uint8_t *st = NULL;
    if (strand) {
        size_t nk = sk->len - (size_t)k + 1;
        st = (uint8_t *)malloc(nk);
        if (!st) { orc_sketch_free(sk); return ORC_ERR_NOMEM; }
        orc_nthi h;
        orc_nthi_init(&h, sk->S, sk->len, (unsigned)k);
        uint64_t c;
        int s1;
        size_t i = 0;
        while (orc_nthi_next(&h, 1, &c, &s1)) st[i++] = (uint8_t)s1;
    }
    while (orc_sketch_next(sk, &code)) {
        if (hash || pos || strand) {
            if ((size_t)n >= cap) { free(st); orc_sketch_free(sk); return ORC_ERR_CAPACITY; }
            long long ix = orc_sketch_index(sk);
            if (hash) hash[n] = code;
            if (pos) pos[n] = (uint32_t)ix;
            if (strand) strand[n] = st[ix];
        }
        n++;
    }
    if (flags) *flags = orc_sketch_flags(sk);
    free(st);
    orc_sketch_free(sk);
    return n;
}

long long orc_minimizer_all(const uint8_t *seq, size_t len, int k, int w, int circular,
                            uint64_t *hash, uint32_t *pos, uint8_t *strand, size_t cap,
                            unsigned *flags) {
    orc_sketch *sk;
    int rc = orc_minimizer_new(seq, len, k, w, circular, &sk);
    if (rc != ORC_OK) return rc;
    return drain(sk, k, hash, pos, strand, cap, flags);
}

long long orc_syncmer_all(const uint8_t *seq, size_t len, int k, int s, int circular,
                          uint64_t *hash, uint32_t *pos, uint8_t *strand, size_t cap,
                          unsigned *flags) {
    orc_sketch *sk;
    int rc = orc_syncmer_new(seq, len, k, s, circular, &sk);
    if (rc != ORC_OK) return rc;
    return drain(sk, k, hash, pos, strand, cap, flags);
}

/* =====================================================================
 * Closed forms (SURVEY.md 8a note V).  Independent second derivation used to
 * cross-check the state machines above and to define what the GPU kernels
 * compute: per window the LEFTMOST argmin, emit when the argmin position
 * changes.
 * ===================================================================== */
typedef struct {
    uint64_t *h;
    uint8_t *st;
    size_t n;
} hvec;

static int hash_vec(const uint8_t *s, size_t L, int k, hvec *v) {
    v->n = L - (size_t)k + 1;
    v->h = (uint64_t *)malloc(v->n * sizeof(uint64_t));
    v->st = (uint8_t *)malloc(v->n);
    if (!v->h || !v->st) { free(v->h); free(v->st); return 0; }
    orc_nthi h;
    orc_nthi_init(&h, s, L, (unsigned)k);
    uint64_t c;
    int s1;
    size_t i = 0;
    while (orc_nthi_next(&h, 1, &c, &s1)) {
        v->h[i] = c;
        v->st[i] = (uint8_t)s1;
        i++;
    }
    return 1;
}

/* ORC_FLAG_FIRST_WINDOW_TIE over the first sorted window h[0..n) (sketch.go:236,351; sketch-protein.go:137):
 * two equal hashes h[t1] == h[t2], t1 < t2 < n, with nothing smaller in (t1, n).  Only such a pair can sit at buf[0]
 * together: two tied entries are both in the buffer while their value is its minimum only if no entry after t1 inside
 * the first window is smaller (later entries are inserted behind equal ones, sketch.go:263-295, so they never reorder).
 * Any other tie never reaches buf[0] with both entries present, so the unstable sort cannot show in the output. */
static void tie_flag(const uint64_t *h, size_t n, unsigned *flags) {
    for (size_t a = 0; a < n; a++)
        for (size_t b = a + 1; b < n; b++) {
            if (h[a] != h[b]) continue;
            int smaller = 0;
            for (size_t c = a + 1; c < n; c++)
                if (h[c] < h[a]) { smaller = 1; break; }
            if (!smaller) { *flags |= ORC_FLAG_FIRST_WINDOW_TIE; return; }
        }
}

static size_t leftmost_argmin(const uint64_t *h, size_t lo, size_t n) {
    size_t p = lo;
    for (size_t q = lo + 1; q < lo + n; q++)
        if (h[q] < h[p]) p = q;
    return p;
}

long long orc_minimizer_closed(const uint8_t *seq, size_t len, int k, int w, int circular,
                               uint64_t *hash, uint32_t *pos, uint8_t *strand, size_t cap,
                               unsigned *flags) {
    if (k < 1) return ORC_ERR_INVALID_K;
    if (w < 1) return ORC_ERR_INVALID_W;
    if (len < (size_t)k + (size_t)w - 1) return ORC_ERR_SHORT_SEQ;
    size_t L;
    uint8_t *s = dup_seq(seq, len, k, circular, &L);
    if (!s) return ORC_ERR_NOMEM;
    hvec v;
    if (!hash_vec(s, L, k, &v)) { free(s); return ORC_ERR_NOMEM; }
    unsigned fl = 0;
    scan_non_acgt(s, L, &fl);
    if (w > 1) tie_flag(v.h, (size_t)w, &fl);
    long long n = 0;
    long long prev = -1;
    for (size_t j = 0; j + (size_t)w <= v.n; j++) {
        size_t p = leftmost_argmin(v.h, j, (size_t)w);
        if ((long long)p == prev) continue;
        prev = (long long)p;
        if ((size_t)n >= cap) { n = ORC_ERR_CAPACITY; break; }
        if (hash) hash[n] = v.h[p];
        if (pos) pos[n] = (uint32_t)p;
        if (strand) strand[n] = v.st[p];
        n++;
    }
    if (flags) *flags = fl;
    free(v.h);
    free(v.st);
    free(s);
    return n;
}

long long orc_syncmer_closed(const uint8_t *seq, size_t len, int k, int sm, int circular,
                             uint64_t *hash, uint32_t *pos, uint8_t *strand, size_t cap,
                             unsigned *flags) {
    if (k < 1) return ORC_ERR_INVALID_K;
    if (sm > k || sm <= 0) return ORC_ERR_INVALID_S;
    if ((long long)len < 2LL * k - sm - 1) return ORC_ERR_SHORT_SEQ;
    size_t L;
    uint8_t *s = dup_seq(seq, len, k, circular, &L);
    if (!s) return ORC_ERR_NOMEM;
    hvec vk, vs;
    if (L < (size_t)k) { free(s); return ORC_ERR_SHORT_SEQ; } /* nthash.NewHasher error, sketch.go:179-182 */
    if (!hash_vec(s, L, k, &vk)) { free(s); return ORC_ERR_NOMEM; }
    unsigned fl = 0;
    scan_non_acgt(s, L, &fl);
    long long n = 0;
    if (sm == k) {
        for (size_t i = 0; i < vk.n; i++) {
            if ((size_t)n >= cap) { n = ORC_ERR_CAPACITY; break; }
            if (hash) hash[n] = vk.h[i];
            if (pos) pos[n] = (uint32_t)i;
            if (strand) strand[n] = vk.st[i];
            n++;
        }
    } else {
        if (!hash_vec(s, L, sm, &vs)) { free(vk.h); free(vk.st); free(s); return ORC_ERR_NOMEM; }
        const long long w = k - sm;
        const long long end = (long long)L - 2LL * k + sm + 1;
        tie_flag(vs.h, (size_t)(2 * w), &fl);
        long long prev = -1;
        for (long long idx = 0; idx <= end; idx++) {
            long long mI = (long long)leftmost_argmin(vs.h, (size_t)idx, (size_t)(2 * w));
            long long b = (mI - idx < w) ? mI : mI - w;
            if (b == prev) continue;
            prev = b;
            if (b > end) continue; /* never reached by idx: dropped (sketch.go:314) */
            if ((size_t)n >= cap) { n = ORC_ERR_CAPACITY; break; }
            if (hash) hash[n] = vk.h[b];
            if (pos) pos[n] = (uint32_t)b;
            if (strand) strand[n] = vk.st[b];
            n++;
        }
        free(vs.h);
        free(vs.st);
    }
    if (flags) *flags = fl;
    free(vk.h);
    free(vk.st);
    free(s);
    return n;
}

/* =====================================================================
 * wyhash.  github.com/zeebo/wyhash v0.0.1 (go.mod:15) is NOT vendored and no
 * reference test checks a protein hash value -> PARITY UNPINNED.  Restated from
 * the published wyhash "version 1" (Wang Yi, 2019-03) that zeebo/wyhash v0.0.1
 * ports: 32-byte rounds, tail switch on len&31, final mum(seed, len ^ p5).
 * Call sites: iterator-protein.go:87, sketch-protein.go:117 (seed = 1).
 * ===================================================================== */
#define WYP0 0xa0761d6478bd642fULL
#define WYP1 0xe7037ed1a0b428dbULL
#define WYP2 0x8ebc6af09c88c6e3ULL
#define WYP3 0x589965cc75374cc3ULL
#define WYP4 0x1d8e4e27c47d124fULL
#define WYP5 0xeb44accab455d165ULL

static inline uint64_t wymum(uint64_t a, uint64_t b) {
    __uint128_t r = (__uint128_t)a * b;
    return (uint64_t)(r >> 64) ^ (uint64_t)r;
}
static inline uint64_t wyr08(const uint8_t *p) { return p[0]; }
static inline uint64_t wyr16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }
static inline uint64_t wyr32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t wyr64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint64_t wyr64s(const uint8_t *p) { return (wyr32(p) << 32) | wyr32(p + 4); }

uint64_t orc_wyhash(const uint8_t *key, size_t len, uint64_t seed) {
    const uint8_t *p = key;
    size_t i;
    for (i = 0; i + 32 <= len; i += 32, p += 32)
        seed = wymum(seed ^ WYP0, wymum(wyr64(p) ^ WYP1, wyr64(p + 8) ^ WYP2) ^
                                      wymum(wyr64(p + 16) ^ WYP3, wyr64(p + 24) ^ WYP4));
    seed ^= WYP0;
    switch (len & 31) {
    case 0: break;
    case 1: seed = wymum(seed, wyr08(p) ^ WYP1); break;
    case 2: seed = wymum(seed, wyr16(p) ^ WYP1); break;
    case 3: seed = wymum(seed, ((wyr16(p) << 8) | wyr08(p + 2)) ^ WYP1); break;
    case 4: seed = wymum(seed, wyr32(p) ^ WYP1); break;
    case 5: seed = wymum(seed, ((wyr32(p) << 8) | wyr08(p + 4)) ^ WYP1); break;
    case 6: seed = wymum(seed, ((wyr32(p) << 16) | wyr16(p + 4)) ^ WYP1); break;
    case 7: seed = wymum(seed, ((wyr32(p) << 24) | (wyr16(p + 4) << 8) | wyr08(p + 6)) ^ WYP1); break;
    case 8: seed = wymum(seed, wyr64s(p) ^ WYP1); break;
    case 9: seed = wymum(wyr64s(p) ^ seed, wyr08(p + 8) ^ WYP2); break;
    case 10: seed = wymum(wyr64s(p) ^ seed, wyr16(p + 8) ^ WYP2); break;
    case 11: seed = wymum(wyr64s(p) ^ seed, ((wyr16(p + 8) << 8) | wyr08(p + 10)) ^ WYP2); break;
    case 12: seed = wymum(wyr64s(p) ^ seed, wyr32(p + 8) ^ WYP2); break;
    case 13: seed = wymum(wyr64s(p) ^ seed, ((wyr32(p + 8) << 8) | wyr08(p + 12)) ^ WYP2); break;
    case 14: seed = wymum(wyr64s(p) ^ seed, ((wyr32(p + 8) << 16) | wyr16(p + 12)) ^ WYP2); break;
    case 15: seed = wymum(wyr64s(p) ^ seed, ((wyr32(p + 8) << 24) | (wyr16(p + 12) << 8) | wyr08(p + 14)) ^ WYP2); break;
    case 16: seed = wymum(wyr64s(p) ^ seed, wyr64s(p + 8) ^ WYP2); break;
    case 17: seed = wymum(wyr64s(p) ^ seed, wyr64s(p + 8) ^ WYP2) ^ wymum(seed, wyr08(p + 16) ^ WYP3); break;
    case 18: seed = wymum(wyr64s(p) ^ seed, wyr64s(p + 8) ^ WYP2) ^ wymum(seed, wyr16(p + 16) ^ WYP3); break;
    case 19: seed = wymum(wyr64s(p) ^ seed, wyr64s(p + 8) ^ WYP2) ^ wymum(seed, ((wyr16(p + 16) << 8) | wyr08(p + 18)) ^ WYP3); break;
    case 20: seed = wymum(wyr64s(p) ^ seed, wyr64s(p + 8) ^ WYP2) ^ wymum(seed, wyr32(p + 16) ^ WYP3); break;
    case 21: seed = wymum(wyr64s(p) ^ seed, wyr64s(p + 8) ^ WYP2) ^ wymum(seed, ((wyr32(p + 16) << 8) | wyr08(p + 20)) ^ WYP3); break;
    case 22: seed = wymum(wyr64s(p) ^ seed, wyr64s(p + 8) ^ WYP2) ^ wymum(seed, ((wyr32(p + 16) << 16) | wyr16(p + 20)) ^ WYP3); break;
    case 23: seed = wymum(wyr64s(p) ^ seed, wyr64s(p + 8) ^ WYP2) ^ wymum(seed, ((wyr32(p + 16) << 24) | (wyr16(p + 20) << 8) | wyr08(p + 22)) ^ WYP3); break;
    case 24: seed = wymum(wyr64s(p) ^ seed, wyr64s(p + 8) ^ WYP2) ^ wymum(seed, wyr64s(p + 16) ^ WYP3); break;
    case 25: seed = wymum(wyr64s(p) ^ seed, wyr64s(p + 8) ^ WYP2) ^ wymum(wyr64s(p + 16) ^ seed, wyr08(p + 24) ^ WYP4); break;
    case 26: seed = wymum(wyr64s(p) ^ seed, wyr64s(p + 8) ^ WYP2) ^ wymum(wyr64s(p + 16) ^ seed, wyr16(p + 24) ^ WYP4); break;
    case 27: seed = wymum(wyr64s(p) ^ seed, wyr64s(p + 8) ^ WYP2) ^ wymum(wyr64s(p + 16) ^ seed, ((wyr16(p + 24) << 8) | wyr08(p + 26)) ^ WYP4); break;
    case 28: seed = wymum(wyr64s(p) ^ seed, wyr64s(p + 8) ^ WYP2) ^ wymum(wyr64s(p + 16) ^ seed, wyr32(p + 24) ^ WYP4); break;
    case 29: seed = wymum(wyr64s(p) ^ seed, wyr64s(p + 8) ^ WYP2) ^ wymum(wyr64s(p + 16) ^ seed, ((wyr32(p + 24) << 8) | wyr08(p + 28)) ^ WYP4); break;
    case 30: seed = wymum(wyr64s(p) ^ seed, wyr64s(p + 8) ^ WYP2) ^ wymum(wyr64s(p + 16) ^ seed, ((wyr32(p + 24) << 16) | wyr16(p + 28)) ^ WYP4); break;
    case 31: seed = wymum(wyr64s(p) ^ seed, wyr64s(p + 8) ^ WYP2) ^ wymum(wyr64s(p + 16) ^ seed, ((wyr32(p + 24) << 24) | (wyr16(p + 28) << 8) | wyr08(p + 30)) ^ WYP4); break;
    }
    return wymum(seed, (uint64_t)len ^ WYP5);
}

/* ---- A7: NewProteinIterator iterator-protein.go:46-73, Next :76-90 (protein input) ---- */
long long orc_protein_hash_all(const uint8_t *aa, size_t len, int k, uint64_t *out, size_t cap) {
    if (k < 1) return ORC_ERR_INVALID_K;                   /* :47 */
    if (len < (size_t)k * 3) return ORC_ERR_SHORT_SEQ;     /* :50 (checked on the INPUT length) */
    long long n = 0;
    for (size_t idx = 0; idx + (size_t)k <= len; idx++) {  /* end = len-k, :71 */
        if (out) {
            if ((size_t)n >= cap) return ORC_ERR_CAPACITY;
            out[n] = orc_wyhash(aa + idx, (size_t)k, 1);   /* :87 */
        }
        n++;
    }
    return n;
}

/* ---- A8: NewProteinMinimizerSketch sketch-protein.go:62-103, Next :106-210 ---- */
static long long protein_minimizer_core(const uint8_t *aa, size_t len, int k, int w, uint64_t *hash,
                                        uint32_t *pos, size_t cap, unsigned *flags);
long long orc_protein_minimizer_all(const uint8_t *aa, size_t len, int k, int w, uint64_t *hash,
                                    uint32_t *pos, size_t cap, unsigned *flags) {
    if (k < 1) return ORC_ERR_INVALID_K;                             /* :63 */
    if (len < (size_t)k * 3) return ORC_ERR_SHORT_SEQ;               /* :66 */
    if (w < 1) return ORC_ERR_INVALID_W;                             /* :70 */
    if (len < (size_t)k * 3 + (size_t)w - 1) return ORC_ERR_SHORT_SEQ; /* :73 */
    return protein_minimizer_core(aa, len, k, w, hash, pos, cap, flags);
}

static long long protein_minimizer_core(const uint8_t *aa, size_t len, int k, int w, uint64_t *hash,
                                        uint32_t *pos, size_t cap, unsigned *flags) {
    /* same sorted-buffer machine as NextMinimizer, fed by wyhash: reuse orc_sketch fields */
    orc_sketch sk;
    memset(&sk, 0, sizeof sk);
    sk.buf = (orc_idxval *)calloc((size_t)w + 2, sizeof(orc_idxval));
    if (!sk.buf) return ORC_ERR_NOMEM;
    sk.w = w;
    sk.r = w - 1;
    sk.preMinIdx = -1;
    const long long end0 = (long long)len - k; /* :93 */
    long long n = 0;
    int rc = 0;
    for (sk.idx = 0; sk.idx <= end0; sk.idx++) { /* :112 */
        uint64_t code = orc_wyhash(aa + sk.idx, (size_t)k, 1); /* :117 */
        int emit = 0;
        if (w == 1) { /* :119 */
            sk.mI = sk.idx;
            sk.mV = code;
            emit = 1;
        } else if (sk.idx < sk.r) { /* :126 */
            sk.buf[sk.buflen].idx = sk.idx;
            sk.buf[sk.buflen].val = code;
            sk.buflen++;
        } else if (sk.idx == sk.r) { /* :134 */
            sk.buf[sk.buflen].idx = sk.idx;
            sk.buf[sk.buflen].val = code;
            sk.buflen++;
            first_window_sort(sk.buf, sk.buflen, &sk.flags); /* :137 */
            sk.mI = sk.buf[0].idx;
            sk.mV = sk.buf[0].val;
            sk.preMinIdx = sk.mI;
            emit = 1;
        } else {
            buf_evict(&sk, sk.idx - w);      /* :151 */
            buf_insert(&sk, sk.idx, code);   /* :162 */
            if (sk.buf[0].idx != sk.preMinIdx) { /* :199 */
                sk.mI = sk.buf[0].idx;
                sk.mV = sk.buf[0].val;
                sk.preMinIdx = sk.mI;
                emit = 1;
            }
        }
        if (emit) {
            if (hash || pos) {
                if ((size_t)n >= cap) { rc = ORC_ERR_CAPACITY; break; }
                if (hash) hash[n] = sk.mV;
                if (pos) pos[n] = (uint32_t)sk.mI;
            }
            n++;
        }
    }
    if (flags) *flags = sk.flags;
    free(sk.buf);
    return rc ? rc : n;
}

long long orc_protein_minimizer_closed(const uint8_t *aa, size_t len, int k, int w,
                                       uint64_t *hash, uint32_t *pos, size_t cap, unsigned *flags) {
    if (k < 1) return ORC_ERR_INVALID_K;
    if (len < (size_t)k * 3) return ORC_ERR_SHORT_SEQ;
    if (w < 1) return ORC_ERR_INVALID_W;
    if (len < (size_t)k * 3 + (size_t)w - 1) return ORC_ERR_SHORT_SEQ;
    size_t nk = len - (size_t)k + 1;
    uint64_t *h = (uint64_t *)malloc(nk * sizeof(uint64_t));
    if (!h) return ORC_ERR_NOMEM;
    for (size_t i = 0; i < nk; i++) h[i] = orc_wyhash(aa + i, (size_t)k, 1);
    unsigned fl = 0;
    if (w > 1) tie_flag(h, (size_t)w, &fl);
    long long n = 0, prev = -1;
    for (size_t j = 0; j + (size_t)w <= nk; j++) {
        size_t p = leftmost_argmin(h, j, (size_t)w);
        if ((long long)p == prev) continue;
        prev = (long long)p;
        if ((size_t)n >= cap) { n = ORC_ERR_CAPACITY; break; }
        if (hash) hash[n] = h[p];
        if (pos) pos[n] = (uint32_t)p;
        n++;
    }
    if (flags) *flags = fl;
    free(h);
    return n;
}

/* =====================================================================
 * DNA/RNA -> protein translation as NewProteinIterator / NewProteinMinimizerSketch apply it to
 * non-protein input: s.Translate(codonTable, frame, trim=false, clean=false,
 * allowUnknownCodon=true, markInitCodonAsM=false)  (iterator-protein.go:62-67,
 * sketch-protein.go:83-88; seq/seq.go:685-708; seq/codon_tables.go:205-285).
 * Pinned by the vectors of seq/codon_tables_test.go:26-135 (tests/golden/codon_golden.json).
 * ===================================================================== */

/* base2code seq/ambiguous_bases.go:28-67: IUPAC letter -> 4-bit set (A1 C2 G4 T/U8); ' ' '*' '-' -> 0;
 * anything else invalid (-1) */
static int base2code(uint8_t b) {
    switch (b) {
    case 'A': case 'a': return 1;
    case 'C': case 'c': return 2;
    case 'G': case 'g': return 4;
    case 'T': case 't': case 'U': case 'u': return 8;
    case 'N': case 'n': return 15;
    case 'M': case 'm': return 3;
    case 'R': case 'r': return 5;
    case 'W': case 'w': return 9;
    case 'S': case 's': return 6;
    case 'Y': case 'y': return 10;
    case 'K': case 'k': return 12;
    case 'V': case 'v': return 7;
    case 'H': case 'h': return 11;
    case 'D': case 'd': return 13;
    case 'B': case 'b': return 14;
    case ' ': case '*': case '-': return 0;
    default: return -1;
    }
}

/* The NCBI genetic codes (https://www.ncbi.nlm.nih.gov/Taxonomy/Utils/wprintgc.cgi), kept as the standard
 * code plus each table's reassigned codons; codon order T,C,A,G for every base, first base slowest
 * (the layout of the five-line blocks at seq/codon_tables.go:431-621). */
static const char STD_CODE[65] = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
static const struct { int id; const char *diff; } GENETIC_CODES[] = {
    {1, ""}, {2, "TGAW ATAM AGA* AGG*"}, {3, "TGAW CTTT CTCT CTAT CTGT ATAM"}, {4, "TGAW"},
    {5, "TGAW ATAM AGAS AGGS"}, {6, "TAAQ TAGQ"}, {9, "TGAW AAAN AGAS AGGS"}, {10, "TGAC"}, {11, ""},
    {12, "CTGS"}, {13, "TGAW ATAM AGAG AGGG"}, {14, "TAAY TGAW AAAN AGAS AGGS"}, {16, "TAGL"},
    {21, "TGAW ATAM AAAN AGAS AGGS"}, {22, "TCA* TAGL"}, {23, "TTA*"}, {24, "TGAW AGAS AGGK"}, {25, "TGAG"},
    {26, "CTGA"}, {27, "TAAQ TAGQ TGAW"}, {28, "TAAQ TAGQ TGAW"}, {29, "TAAY TAGY"}, {30, "TAAE TAGE"},
    {31, "TAAE TAGE TGAW"},
};

/* 64 amino acids of table `id` in TCAG order; returns 0, or -1 for an unknown table (seq.go:691) */
int orc_genetic_code(int id, char aa64[65]) {
    static const char TCAG[] = "TCAG";
    for (size_t t = 0; t < sizeof GENETIC_CODES / sizeof GENETIC_CODES[0]; t++) {
        if (GENETIC_CODES[t].id != id) continue;
        memcpy(aa64, STD_CODE, 65);
        for (const char *d = GENETIC_CODES[t].diff; *d; d += (d[4] ? 5 : 4)) {
            int idx = 0;
            for (int j = 0; j < 3; j++) idx = idx * 4 + (int)(strchr(TCAG, d[j]) - TCAG);
            aa64[idx] = d[3];
        }
        return 0;
    }
    return -1;
}

/* codonTableFromText seq/codon_tables.go:317-429: 16x16x16 matrix indexed by base2code; the 64 plain
 * codons first (:329-341), then the three passes that give a codon with ambiguity letters an amino acid
 * when the letters it stands for agree (:350-427).  In a pass the groups of different amino acids write
 * disjoint entries, so Go's map iteration order does not matter. */
int orc_codon_matrix(int id, uint8_t m[16][16][16]) {
    static const int CODE_TCAG[4] = {8, 2, 1, 4};
    char aa64[65];
    if (orc_genetic_code(id, aa64) != 0) return -1;
    memset(m, 0, 4096);
    for (int i = 0; i < 64; i++) m[CODE_TCAG[i >> 4]][CODE_TCAG[(i >> 2) & 3]][CODE_TCAG[i & 3]] = (uint8_t)aa64[i];
    for (int pass = 0; pass < 3; pass++) {          /* 0: third base :350, 1: second :376, 2: first :402 */
        for (int a = 1; a < 16; a++) {
            for (int b = 1; b < 16; b++) {
                /* group the running index c by amino acid: set[aa] = OR of the codes that carry it (Codes2AmbCode) */
                int set[256];
                memset(set, 0, sizeof set);
                for (int c = 1; c < 16; c++) {
                    uint8_t aa = pass == 0 ? m[a][b][c] : pass == 1 ? m[a][c][b] : m[c][a][b];
                    if (aa) set[aa] |= c;
                }
                for (int aa = 1; aa < 256; aa++) {
                    if (!set[aa]) continue;
                    /* AmbCodes2Codes seq/ambiguous_bases.go:153-175: every non-empty sub-set of the code */
                    for (int c = 1; c < 16; c++) {
                        if ((c & set[aa]) != c) continue;
                        if (pass == 0) m[a][b][c] = (uint8_t)aa;
                        else if (pass == 1) m[a][c][b] = (uint8_t)aa;
                        else m[c][a][b] = (uint8_t)aa;
                    }
                }
            }
        }
    }
    return 0;
}

/* CodonTable.Get seq/codon_tables.go:157-177 with allowUnknownCodon = true */
static uint8_t codon_get(uint8_t m[16][16][16], uint8_t c0, uint8_t c1, uint8_t c2) {
    int i = base2code(c0), j = base2code(c1), k = base2code(c2);
    if (i < 0 || j < 0 || k < 0) return 'X';                 /* :160-163 */
    if (c0 == '-' && c1 == '-' && c2 == '-') return '-';     /* :167 */
    uint8_t aa = m[i][j][k];
    return aa ? aa : 'X';                                     /* :172-174 */
}

/* DNA.PairLetter seq/alphabet.go:313-325 with the DNA alphabet of :353-359: acgtACGT are complemented;
 * gap and ambiguous letters map to themselves; for any other byte the error is ignored by the caller
 * (codon_tables.go:226-228) and the byte itself is used */
static uint8_t dna_pair_letter(uint8_t b) {
    switch (b) {
    case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
    case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
    default: return b;
    }
}

/* CodonTable.Translate seq/codon_tables.go:205-285 (allowUnknownCodon = true, markInitCodonAsM = false).
 * Returns the number of residues (written to out if non-NULL, at most cap), -1 unknown table,
 * -2 bad frame, -3 sequence shorter than 3 (:206). */
long long orc_translate(const uint8_t *nt, size_t len, int table, int frame, int trim, int clean,
                        uint8_t *out, size_t cap) {
    uint8_t m[16][16][16];
    if (orc_codon_matrix(table, m) != 0) return -1;
    if (len < 3) return -3;
    if (frame < -3 || frame > 3 || frame == 0) return -2;
    long long n = 0;
    if (frame < 0) {
        for (long long i = (long long)len + frame; i >= 2; i -= 3) {                  /* :224 */
            uint8_t aa = codon_get(m, dna_pair_letter(nt[i]), dna_pair_letter(nt[i - 1]), dna_pair_letter(nt[i - 2]));
            if (trim && (aa == 'X' || aa == '*')) break;                              /* :246 */
            if (clean && aa == '*') aa = 'X';
            if (out) { if ((size_t)n >= cap) return ORC_ERR_CAPACITY; out[n] = aa; }
            n++;
        }
    } else {
        for (size_t i = (size_t)frame - 1; i + 2 < len; i += 3) {                     /* :256 */
            uint8_t aa = codon_get(m, nt[i], nt[i + 1], nt[i + 2]);
            if (trim && (aa == 'X' || aa == '*')) break;
            if (clean && aa == '*') aa = 'X';
            if (out) { if ((size_t)n >= cap) return ORC_ERR_CAPACITY; out[n] = aa; }
            n++;
        }
    }
    return n;
}

/* A7 on nucleotide input (iterator-protein.go:46-73): the length check is on the INPUT (:50); the
 * translation may then be shorter than k, in which case end < 0 and Next() ends at once (:81). */
long long orc_protein_hash_nt(const uint8_t *nt, size_t len, int k, int table, int frame, uint64_t *out, size_t cap) {
    if (k < 1) return ORC_ERR_INVALID_K;
    if (len < (size_t)k * 3) return ORC_ERR_SHORT_SEQ;
    uint8_t *aa = (uint8_t *)malloc(len / 3 + 2);
    if (!aa) return ORC_ERR_NOMEM;
    long long P = orc_translate(nt, len, table, frame, 0, 0, aa, len / 3 + 2);
    if (P < 0) { free(aa); return ORC_ERR_ILLEGAL_BASE; }  /* Translate's own errors (table, frame): surfaced as "other" */
    long long n = 0;
    for (long long idx = 0; idx + k <= P; idx++) {
        if (out) {
            if ((size_t)n >= cap) { free(aa); return ORC_ERR_CAPACITY; }
            out[n] = orc_wyhash(aa + idx, (size_t)k, 1);
        }
        n++;
    }
    free(aa);
    return n;
}

/* A8 on nucleotide input (sketch-protein.go:62-103): both length checks are on the INPUT (:66,:73); a
 * translation with fewer than w k-mers never completes a window and yields nothing (:126-131). */
long long orc_protein_minimizer_nt(const uint8_t *nt, size_t len, int k, int w, int table, int frame,
                                   uint64_t *hash, uint32_t *pos, size_t cap, unsigned *flags) {
    if (k < 1) return ORC_ERR_INVALID_K;
    if (len < (size_t)k * 3) return ORC_ERR_SHORT_SEQ;
    if (w < 1) return ORC_ERR_INVALID_W;
    if (len < (size_t)k * 3 + (size_t)w - 1) return ORC_ERR_SHORT_SEQ;
    uint8_t *aa = (uint8_t *)malloc(len / 3 + 2);
    if (!aa) return ORC_ERR_NOMEM;
    long long P = orc_translate(nt, len, table, frame, 0, 0, aa, len / 3 + 2);
    if (P < 0) { free(aa); return ORC_ERR_ILLEGAL_BASE; }
    long long n = protein_minimizer_core(aa, (size_t)P, k, w, hash, pos, cap, flags);
    free(aa);
    return n;
}

/* =====================================================================
 * Batch driver for bench.py's cpu_baseline leg: one iterator per read, as a
 * Go caller would, OpenMP over reads.  Uses the STATE MACHINES (same per-element
 * work and asymptotics as the Go code), not the closed forms.
 * ===================================================================== */
int orc_batch_run(int kind, const uint8_t *seqs, const uint64_t *offsets, uint32_t n, int k,
                  int w_or_s, int threads, uint64_t *n_tuples, uint64_t *checksum) {
    uint64_t tot = 0, sum = 0;
    int err = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads) reduction(+ : tot, sum)
#endif
  {
    /* one recycled sketch object per thread: what sync.Pool amounts to for a worker goroutine that finishes one iterator
     * before it asks for the next (sketch.go:79,214,321); the reads are borrowed, not copied (sketch.go:106-110) */
    orc_sketch *pooled = NULL;
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (long long r = 0; r < (long long)n; r++) {
        const uint8_t *s = seqs + offsets[r];
        size_t len = (size_t)(offsets[r + 1] - offsets[r]);
        uint64_t code;
        if (kind == 2) {
            orc_nthi h;
            if (k < 1 || len < (size_t)k) continue;
            orc_nthi_init(&h, s, len, (unsigned)k);
            uint64_t i = 0;
            while (orc_nthi_next(&h, 1, &code, NULL)) {
                sum += code * (2 * i + 1);
                i++;
                tot++;
            }
        } else if (kind == 4 || kind == 5) {
            orc_sketch *sk;
            int rc = (kind == 4) ? minimizer_init(pooled, 1, s, len, k, w_or_s, 0, &sk)
                                 : syncmer_init(pooled, 1, s, len, k, w_or_s, 0, &sk);
            if (rc == ORC_ERR_NOMEM) pooled = NULL; /* sketch_alloc released it */
            if (rc != ORC_OK) continue;
            while (orc_sketch_next(sk, &code)) {
                sum += code * (2 * (uint64_t)orc_sketch_index(sk) + 1);
                tot++;
            }
            pooled = sk; /* back to the pool */
        } else if (kind == 7) {
            enum { CAP = 4096 };
            uint64_t hb[CAP];
            uint32_t pb[CAP];
            long long c = orc_protein_minimizer_all(s, len, k, w_or_s, hb, pb, CAP, NULL);
            for (long long i = 0; i < c; i++) sum += hb[i] * (2 * (uint64_t)pb[i] + 1);
            if (c > 0) tot += (uint64_t)c;
        } else if (kind == 1 || kind == 3 || kind == 6) { /* every-position kinds: k-mer codes, SimHash (m=5, scale=5), protein hash */
            enum { CAP2 = 8192 };
            uint64_t hb[CAP2];
            long long c = kind == 1 ? orc_kmer_all(s, len, k, 1, 0, hb, CAP2)
                        : kind == 3 ? orc_simhash_all(s, len, k, 5, 5, 1, 0, hb, CAP2)
                                    : orc_protein_hash_all(s, len, k, hb, CAP2);
            for (long long i = 0; i < c; i++) sum += hb[i] * (2 * (uint64_t)i + 1);
            if (c > 0) tot += (uint64_t)c;
        } else {
            err = 1;
        }
    }
    orc_sketch_free(pooled);
  }
    (void)threads;
    if (n_tuples) *n_tuples = tot;
    if (checksum) *checksum = sum;
    return err ? -1 : 0;
}
