/*
 * bio_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the `sketches/` hot path of shenwei356/bio
 * (reference @ v0.13.8, /root/reference).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may link or call this.  The product path
 * (bio_amd/csrc, libbiosketch.so) never includes or links it.
 *
 * Parity status (see DESIGN.md "Oracle pinning"):
 *   - ntHash arithmetic .......... PINNED by sketches/sketch_test.go:67-72
 *                                  (5 known-answer hashes, k=5 w=3).
 *   - minimizer / syncmer logic ... restated line by line from
 *                                  sketches/sketch.go:205-477; closed forms
 *                                  cross-checked against the state machines.
 *   - first-window tie order ...... PARITY UNPINNED (twotwotwo/sorts Quicksort
 *                                  is unstable and un-vendored); the oracle
 *                                  uses a stable sort and raises
 *                                  ORC_FLAG_FIRST_WINDOW_TIE.
 *   - non-ACGT bytes, k > 64 ...... PARITY UNPINNED (will-rowe/nthash v0.4.0
 *                                  un-vendored; published ntHash-1 table used).
 *   - wyhash (protein paths) ...... PARITY UNPINNED (zeebo/wyhash v0.0.1
 *                                  un-vendored; published wyhash-v1 restated).
 */
#ifndef BIO_ORACLE_H
#define BIO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Error codes: one per sentinel error of the reference
 * (sketches/iterator.go:34-53, sketches/sketch.go:32-42).  Negative on purpose:
 * the *_all() helpers return a count >= 0 or one of these. */
enum {
    ORC_OK = 0,
    ORC_ERR_INVALID_K = -1,     /* ErrInvalidK     iterator.go:34 */
    ORC_ERR_EMPTY_SEQ = -2,     /* ErrEmptySeq     iterator.go:37 */
    ORC_ERR_SHORT_SEQ = -3,     /* ErrShortSeq     iterator.go:40 */
    ORC_ERR_ILLEGAL_BASE = -4,  /* ErrIllegalBase  iterator.go:43 */
    ORC_ERR_K_TOO_LARGE = -5,   /* ErrKTooLarge    iterator.go:46 */
    ORC_ERR_INVALID_M = -6,     /* ErrInvalidM     iterator.go:49 */
    ORC_ERR_INVALID_SCALE = -7, /* ErrInvalidScale iterator.go:52 */
    ORC_ERR_INVALID_S = -8,     /* ErrInvalidS     sketch.go:32 */
    ORC_ERR_INVALID_W = -9,     /* ErrInvalidW     sketch.go:35 */
    ORC_ERR_NOMEM = -100,
    ORC_ERR_CAPACITY = -101
};

/* per-sequence flags (same bit values as include/biosketch.h BSK_ST_*) */
enum {
    ORC_FLAG_FIRST_WINDOW_TIE = 0x10, /* a tied pair in the first sorted window with nothing smaller behind its first entry */
    ORC_FLAG_HAS_NON_ACGT = 0x20      /* a byte outside ACGTacgt was hashed */
};

/* ---- ntHash-1 (will-rowe/nthash v0.4.0 as called from iterator.go:649,659) ---- */
uint64_t orc_seed_fwd(uint8_t b);
uint64_t orc_seed_rev(uint8_t b); /* seed of the complement: table[b & 0x07] */

typedef struct {
    const uint8_t *seq;
    size_t len;
    unsigned k;
    uint64_t fh, rh;
    size_t cur, max_idx;
} orc_nthi;

int orc_nthi_init(orc_nthi *h, const uint8_t *seq, size_t len, unsigned k);
/* returns 1 and writes *hash (and *strand: 1 iff the reverse hash was chosen) or 0 at end */
int orc_nthi_next(orc_nthi *h, int canonical, uint64_t *hash, int *strand);

/* ---- A2: NewHashIterator / NextHash  (iterator.go:615-665) ----
 * out/strand may be NULL (count only).  Returns #hashes or ORC_ERR_*. */
long long orc_nthash_all(const uint8_t *seq, size_t len, int k, int canonical, int circular,
                         uint64_t *out, uint8_t *strand, size_t cap);

/* ---- A1: NewKmerIterator / NextKmer  (iterator.go:668-759) ----
 * canonical=0 emits the forward strand then the reverse-complement strand
 * (iterator.go:713-723) => 2*(L-k+1) codes. */
long long orc_kmer_all(const uint8_t *seq, size_t len, int k, int canonical, int circular,
                       uint64_t *out, size_t cap);
/* the same for a sequence whose Alphabet is not DNAredundant (numbered as bsk_alphabet: 2 DNA, 3 RNA, 4 RNAredundant,
 * 5 Unlimit): only the second strand of canonical = 0 differs (RevComInplace pairs letters per alphabet) */
long long orc_kmer_all_alpha(const uint8_t *seq, size_t len, int k, int canonical, int circular, int alphabet,
                             uint64_t *out, size_t cap);

/* ---- A3: NewSimHashIterator / NextSimHash  (iterator.go:113-612) ---- */
long long orc_simhash_all(const uint8_t *seq, size_t len, int k, int m, int scale, int canonical,
                          int circular, uint64_t *out, size_t cap);

/* ---- A5/A6: Sketch (sketch.go) -- line-by-line state machine ---- */
typedef struct {
    long long idx;
    uint64_t val;
} orc_idxval; /* IdxValue sketch.go:496 */

typedef struct orc_sketch orc_sketch;
/* constructors return ORC_OK or an error; *out owns a private copy of seq */
int orc_minimizer_new(const uint8_t *seq, size_t len, int k, int w, int circular, orc_sketch **out);
int orc_syncmer_new(const uint8_t *seq, size_t len, int k, int s, int circular, orc_sketch **out);
int orc_sketch_next(orc_sketch *s, uint64_t *code); /* Next()  sketch.go:480 -> 1/0 */
long long orc_sketch_index(const orc_sketch *s);   /* Index() sketch.go:488 */
unsigned orc_sketch_flags(const orc_sketch *s);
void orc_sketch_free(orc_sketch *s);

/* state machine drained into arrays; pos[i] = Index() after each Next();
 * strand[i] = strand of the canonical k-mer at that index.  Returns count or ORC_ERR_*. */
long long orc_minimizer_all(const uint8_t *seq, size_t len, int k, int w, int circular,
                            uint64_t *hash, uint32_t *pos, uint8_t *strand, size_t cap,
                            unsigned *flags);
long long orc_syncmer_all(const uint8_t *seq, size_t len, int k, int s, int circular,
                          uint64_t *hash, uint32_t *pos, uint8_t *strand, size_t cap,
                          unsigned *flags);

/* closed forms (SURVEY.md section 8a note V): leftmost argmin per window + position dedup */
long long orc_minimizer_closed(const uint8_t *seq, size_t len, int k, int w, int circular,
                               uint64_t *hash, uint32_t *pos, uint8_t *strand, size_t cap,
                               unsigned *flags);
long long orc_syncmer_closed(const uint8_t *seq, size_t len, int k, int s, int circular,
                             uint64_t *hash, uint32_t *pos, uint8_t *strand, size_t cap,
                             unsigned *flags);

/* ---- A7/A8: protein (iterator-protein.go, sketch-protein.go); input is amino acids ---- */
uint64_t orc_wyhash(const uint8_t *p, size_t len, uint64_t seed);
long long orc_protein_hash_all(const uint8_t *aa, size_t len, int k, uint64_t *out, size_t cap);
long long orc_protein_minimizer_all(const uint8_t *aa, size_t len, int k, int w, uint64_t *hash,
                                    uint32_t *pos, size_t cap, unsigned *flags);
long long orc_protein_minimizer_closed(const uint8_t *aa, size_t len, int k, int w,
                                       uint64_t *hash, uint32_t *pos, size_t cap, unsigned *flags);

/* ---- A7/A8 on DNA/RNA input: Translate(table, frame, trim=false, clean=false, allowUnknownCodon=true,
 * markInitCodonAsM=false) first (seq/codon_tables.go:205-285), length checks on the nucleotide length ---- */
int orc_genetic_code(int id, char aa64[65]);
int orc_codon_matrix(int id, uint8_t m[16][16][16]);
long long orc_translate(const uint8_t *nt, size_t len, int table, int frame, int trim, int clean,
                        uint8_t *out, size_t cap);
long long orc_protein_hash_nt(const uint8_t *nt, size_t len, int k, int table, int frame, uint64_t *out, size_t cap);
long long orc_protein_minimizer_nt(const uint8_t *nt, size_t len, int k, int w, int table, int frame,
                                   uint64_t *hash, uint32_t *pos, size_t cap, unsigned *flags);

/* ---- batch drivers (cpu_baseline leg of bench.py; OpenMP over reads) ----
 * kind: 2 = ntHash stream, 4 = minimizer, 5 = syncmer, 7 = protein minimizer.
 * seqs: concatenated bytes, offsets[n+1].  Only tuple counts and an
 * order-independent checksum are returned (sum over tuples of hash*(2*pos+1)). */
int orc_batch_run(int kind, const uint8_t *seqs, const uint64_t *offsets, uint32_t n, int k,
                  int w_or_s, int threads, uint64_t *n_tuples, uint64_t *checksum);

#ifdef __cplusplus
}
#endif
#endif
