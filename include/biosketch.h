/*
 * biosketch.h -- C ABI of libbiosketch.so, the MI355X (gfx950) k-mer sketching engine.
 *
 * This is the drop-in boundary for the `sketches/` package of shenwei356/bio
 * (reference v0.13.8; file:line below are relative to the reference root).  The
 * reference exposes per-sequence pull iterators; a device cannot be driven one
 * value at a time, so the boundary is "batch in -> all tuples out", and the
 * reference's iterator types become cursors over one read's slice of a result
 * (bindings/go/sketches, bio_amd/csrc/sketches.hpp, bio_amd/sketches.py).
 *
 *   reference constructor / pull method                     ->  bsk_params.kind
 *   NewKmerIterator / NextKmer        iterator.go:668,708   ->  BSK_KMER
 *   NewHashIterator / NextHash        iterator.go:615,658   ->  BSK_NTHASH
 *   NewSimHashIterator / NextSimHash  iterator.go:113,191   ->  BSK_SIMHASH
 *   NewMinimizerSketch / NextMinimizer sketch.go:85,205     ->  BSK_MINIMIZER
 *   NewSyncmerSketch / NextSyncmer    sketch.go:142,312     ->  BSK_SYNCMER
 *   NewProteinIterator / Next         iterator-protein.go:46,76      -> BSK_PROT_HASH
 *   NewProteinMinimizerSketch / Next  sketch-protein.go:62,106       -> BSK_PROT_MINIMIZER
 *   Index()  iterator.go:776 sketch.go:488 iterator-protein.go:93 sketch-protein.go:213
 *                                                           ->  bsk_result pos[] (bit 31 = strand)
 *
 * Plain C: pointers and sizes only, no C++/torch types.  All functions return
 * BSK_OK (0) or a bsk_err; nothing aborts or throws across the boundary.  Host
 * pointers passed in are never retained after the call returns (cgo rule).
 * One bsk_ctx = one GPU + one HIP stream; contexts are independent and may be
 * used from different threads concurrently; a single context is not re-entrant.
 */
#ifndef BIOSKETCH_H
#define BIOSKETCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BSK_ABI_VERSION 1

/* Call-level errors.  1..11 are the reference's sentinel errors, returned when a
 * constructor argument is invalid for EVERY read of the batch
 * (iterator.go:34-53, sketch.go:32-42). */
typedef enum bsk_err {
    BSK_OK = 0,
    BSK_ERR_INVALID_K = 1,     /* ErrInvalidK      iterator.go:34  k < 1 */
    BSK_ERR_EMPTY_SEQ = 2,     /* ErrEmptySeq      iterator.go:37  (declared, never returned upstream) */
    BSK_ERR_SHORT_SEQ = 3,     /* ErrShortSeq      iterator.go:40  (per read: see BSK_ST_SHORT) */
    BSK_ERR_ILLEGAL_BASE = 4,  /* ErrIllegalBase   iterator.go:43  (per read: see BSK_ST_ILLEGAL) */
    BSK_ERR_K_TOO_LARGE = 5,   /* ErrKTooLarge     iterator.go:46 */
    BSK_ERR_INVALID_M = 6,     /* ErrInvalidM      iterator.go:49 */
    BSK_ERR_INVALID_SCALE = 7, /* ErrInvalidScale  iterator.go:52 */
    BSK_ERR_INVALID_S = 8,     /* ErrInvalidS      sketch.go:32 */
    BSK_ERR_INVALID_W = 9,     /* ErrInvalidW      sketch.go:35 */
    BSK_ERR_BUF_NIL = 10,      /* ErrBufNil        sketch.go:38 (declared, never returned upstream) */
    BSK_ERR_BUF_NOT_EMPTY = 11,/* ErrBufNotEmpty   sketch.go:41 (declared, never returned upstream) */
    BSK_ERR_ARG = 64,          /* NULL pointer / inconsistent sizes */
    BSK_ERR_NOMEM = 65,        /* host or device allocation failed */
    BSK_ERR_DEVICE = 66,       /* HIP runtime error; text in bsk_last_error() */
    BSK_ERR_UNSUPPORTED = 67,  /* valid upstream, not implemented here (see DESIGN.md scope) */
    BSK_ERR_NO_DEVICE = 68,    /* no gfx950 device visible: there is NO CPU fallback */
    BSK_ERR_IO = 69,           /* FASTA/Q reader: cannot open / read */
    BSK_ERR_NOT_FASTX = 70,    /* ErrNotFASTXFormat seqio/fastx/reader.go:16 */
    BSK_ERR_BAD_FASTQ = 71,    /* ErrBadFASTQFormat / ErrUnequalSeqAndQual reader.go:19-22 */
    BSK_ERR_STOPPED = 72       /* bsk_pipeline_run: on_chunk returned non-zero and the run was stopped on the consumer's request */
} bsk_err;

typedef enum bsk_kind {
    BSK_KMER = 1,
    BSK_NTHASH = 2,
    BSK_SIMHASH = 3,
    BSK_MINIMIZER = 4,
    BSK_SYNCMER = 5,
    BSK_PROT_HASH = 6,
    BSK_PROT_MINIMIZER = 7
} bsk_kind;

typedef enum bsk_alphabet {
    BSK_ALPHA_DNA = 0,    /* nucleotides (seq.DNAredundant): hashed with ntHash, 2-bit packed on device */
    BSK_ALPHA_PROTEIN = 1,/* seq.Protein: one byte per residue (the `S.Alphabet == seq.Protein`
                             branch of iterator-protein.go:62 / sketch-protein.go:83) */
    /* The other nucleotide alphabets of seq/alphabet.go:352-383.  They are BSK_ALPHA_DNA everywhere but in ONE place: the
     * second strand of NextKmer's two-strand mode comes from RevComInplace (iterator.go:719), which pairs letters with the
     * sequence's OWN alphabet and leaves every other byte as it is (seq/seq.go:386-392):
     *   DNA            acgtACGT <-> tgcaTGCA            RNA            acguACGU <-> ugcaUGCA  (a 'T' stays)
     *   DNAredundant   + ryswkmbdhv <-> yrswmkvhdb      RNAredundant   + ryswkmbdhv <-> yrswmkvhdb
     *   Unlimit        not complemented at all (seq/seq.go:381-383): the second strand is the reversed sequence. */
    BSK_ALPHA_DNA_PLAIN = 2,
    BSK_ALPHA_RNA = 3,
    BSK_ALPHA_RNA_REDUNDANT = 4,
    BSK_ALPHA_UNLIMIT = 5
} bsk_alphabet;

/* Per-read status byte (bsk_result status[]).  Low nibble = what the reference
 * constructor / iterator would have reported for that read; high nibble = flags. */
#define BSK_ST_OK 0x00
#define BSK_ST_SHORT 0x01             /* constructor would return ErrShortSeq; 0 tuples */
#define BSK_ST_ILLEGAL 0x02           /* NextKmer hit ErrIllegalBase (iterator.go:731,746); tuples before it kept */
#define BSK_ST_CODE_MASK 0x0f
#define BSK_ST_FIRST_WINDOW_TIE 0x10  /* the first sorted window (first w k-mer hashes; first 2(k-s) s-mer hashes for
                                         syncmers) holds two equal hashes h[t1] == h[t2], t1 < t2, with nothing smaller
                                         behind t1 inside that window -- the only ties that can sit at buf[0] together,
                                         where upstream's unstable sorts.Quicksort (sketch.go:236,351) may order them
                                         either way; this engine returns the leftmost.  Bit-exactness vs Go is claimed
                                         for reads WITHOUT this flag.  (A tie with a smaller hash behind its first entry
                                         never reaches the front of the buffer while both entries are in it.) */
#define BSK_ST_HAS_NON_ACGT 0x20      /* a byte outside ACGTacgt was hashed (ntHash seed table for such
                                         bytes is unpinned upstream) */

/* reference words of a result (bsk_result_device): (first_tuple << 24) | n_tuples, bit 63 = the read's tuples are 64 apart */
#define BSK_REF_ROWS (1ULL << 63)
#define BSK_REF_FIRST(ref) (((ref) & ~BSK_REF_ROWS) >> 24)
#define BSK_REF_COUNT(ref) ((ref) & 0xffffffULL)
#define BSK_REF_STRIDE(ref) (((ref) & BSK_REF_ROWS) ? 64ULL : 1ULL)

#define BSK_POS_STRAND_BIT 0x80000000u /* pos[] bit 31: 1 iff the reverse-strand hash was the canonical one */
#define BSK_POS_MASK 0x7fffffffu

/* Mirrors the reference constructor arguments 1:1. */
typedef struct bsk_params {
    int32_t kind;       /* bsk_kind */
    int32_t k;          /* k-mer size (all kinds) */
    int32_t w;          /* MINIMIZER / PROT_MINIMIZER: window, sketch.go:85 `w`, sketch-protein.go:62 `w` */
    int32_t s;          /* SYNCMER: s-mer size, sketch.go:142 `s` */
    int32_t m;          /* SIMHASH: m-mer size, iterator.go:113 `m` */
    int32_t scale;      /* SIMHASH: FracMinHash scale, iterator.go:113 `scale` */
    int32_t canonical;  /* KMER / NTHASH / SIMHASH: `canonical` argument (sketches are always canonical) */
    int32_t circular;   /* `circular` argument: first k-1 bases appended (iterator.go:642-646) */
    int32_t codon_table;/* PROT_* on a DNA batch: NCBI genetic-code id for the translation (ignored for protein batches) */
    int32_t frame;      /* PROT_* on a DNA batch: 1,2,3,-1,-2,-3 */
} bsk_params;

typedef struct bsk_ctx bsk_ctx;
typedef struct bsk_batch bsk_batch;
typedef struct bsk_result bsk_result;

/* ---- context ---------------------------------------------------------------- */
int bsk_abi_version(void);
int bsk_device_count(int *n);                      /* number of visible gfx950 devices */
int bsk_ctx_create(int device, bsk_ctx **ctx);     /* hipSetDevice + one stream */
void bsk_ctx_destroy(bsk_ctx *ctx);
int bsk_ctx_sync(bsk_ctx *ctx);                    /* wait for the context's stream */
/* The library's developer switches (environment variables BSK_*, DESIGN.md "Switches") are read ONCE, by bsk_ctx_create -- never on
 * the bsk_sketch path.  A process that changes them afterwards (the test suite does) calls this to have the context read them again. */
int bsk_ctx_reload_options(bsk_ctx *ctx);
int bsk_build_has_experiments(void);               /* 1 iff built with `make EXPERIMENTS=1`: the measured-and-rejected kernels behind BSK_SEG / BSK_WPR are in */
const char *bsk_last_error(const bsk_ctx *ctx);    /* text of the last BSK_ERR_DEVICE/ARG on this ctx */
const char *bsk_err_name(int err);                 /* "ErrShortSeq", ... (the reference's names) */

/* ---- batches: what a Go caller collects from fastx.Record.Seq.Seq ---------------
 * seqio/fastx/reader.go:229-232 reuses the record buffer, so the shim copies each
 * record's bytes into one contiguous `bytes` array with `offsets[n+1]`.  The call
 * copies to the device and (DNA) packs to 2 bits/base there; host memory is not
 * referenced after return. */
int bsk_batch_from_ascii(bsk_ctx *ctx, const uint8_t *bytes, const uint64_t *offsets, uint64_t n,
                         int alphabet, bsk_batch **out);
/* Pre-packed DNA: words[] holds 16 bases per uint32 (base i of a read in bits
 * [2*(i%16), 2*(i%16)+2) of word i/16, A0 C1 G2 T3); each read starts on a word
 * boundary; desc[r] = (first_word << 24) | n_bases.  n_words = total words. */
int bsk_batch_from_packed(bsk_ctx *ctx, const uint32_t *words, uint64_t n_words,
                          const uint64_t *desc, uint64_t n, bsk_batch **out);
/* Synthetic i.i.d. uniform reads generated ON DEVICE (bench / full-size tests):
 * DNA over ACGT (packed) or protein over ACDEFGHIKLMNPQRSTVWY; counter-based
 * splitmix64(seed, index), so any shard is reproducible. */
int bsk_batch_synth(bsk_ctx *ctx, int alphabet, uint64_t n, uint32_t len, uint64_t seed,
                    bsk_batch **out);
int bsk_batch_info(const bsk_batch *b, uint64_t *n_reads, uint64_t *n_bases, uint64_t *device_bytes,
                   uint64_t *n_non_acgt_reads);
/* Decode reads [first, first+count) back to ASCII on the host (tests: feed the oracle
 * the exact bytes the device hashed).  bytes_cap = capacity of bytes[]; offsets[count+1]. */
int bsk_batch_fetch_ascii(bsk_ctx *ctx, const bsk_batch *b, uint64_t first, uint64_t count,
                          uint8_t *bytes, uint64_t bytes_cap, uint64_t *offsets);
/* DNA/RNA batch -> protein batch on the device: (*seq.Seq).Translate(codon_table, frame, trim=false, clean=false,
 * allowUnknownCodon=true, markInitCodonAsM=false), seq/seq.go:685-708 + seq/codon_tables.go:205-285 -- the call that
 * NewProteinIterator (iterator-protein.go:62-67) and NewProteinMinimizerSketch (sketch-protein.go:83-88) make for
 * non-protein input.  frame: 1,2,3,-1,-2,-3; codon_table: an NCBI genetic-code id of seq/codon_tables.go:431-621.
 * bsk_sketch does this internally when a PROT_* kind is run on a DNA batch (and then applies the constructors'
 * length checks to the NUCLEOTIDE length, as the reference does); this entry point exposes the translation itself. */
int bsk_batch_translate(bsk_ctx *ctx, const bsk_batch *dna, int codon_table, int frame, bsk_batch **out);
/* Host-only: the tables the translate kernel uses for `codon_table` -- [0,4096) amino acid of the codon whose bases
 * are the IUPAC sets (i,j,k) at index i*256+j*16+k ('X' where the reference's 16x16x16 matrix is empty,
 * codon_tables.go:172-174,317-429), [4096,4352) letter -> set (16 = not a base letter, ambiguous_bases.go:28-67),
 * [4352,4416) the 64 plain codons by 2-bit code (A0 C1 G2 T3, first base most significant).  lut_bytes >= 4416. */
int bsk_codon_lut(int codon_table, uint8_t *lut, uint64_t lut_bytes);
void bsk_batch_destroy(bsk_batch *b);

/* ---- FASTA/FASTQ feeding (host side) --------------------------------
 * Record reader with the semantics of seqio/fastx Reader.Read / parseRecord (reader.go:233-471): format from the first
 * non-newline byte, records start at '>' / '@' after a newline, multi-line FASTA and FASTQ, CR dropped, quality lines that
 * start with '@', the reference's edge cases (record without sequence, empty file, newline-only file).  gzip or plain,
 * "-" = stdin.  bsk_fastx_read_chunk returns up to max_records records / stops after max_bytes sequence bytes (0 = no
 * limit; at least one record): concatenated sequence bytes + offsets[n+1], header lines + offsets[n+1], and for FASTQ the
 * concatenated qualities (same offsets as the sequences) -- pointers into reader-owned memory, valid until the next
 * call.  *n == 0 with BSK_OK = end of file.  Errors: BSK_ERR_NOT_FASTX (ErrNotFASTXFormat), BSK_ERR_BAD_FASTQ
 * (ErrBadFASTQFormat / ErrUnequalSeqAndQual), BSK_ERR_IO.  bsk_fastx_info: is_fastq (-1 before the first record) and the
 * alphabet guessed from the first sequence as the reference does, in its order (seq/alphabet.go:413-452): BSK_ALPHA_DNA_PLAIN,
 * BSK_ALPHA_RNA, BSK_ALPHA_DNA (DNAredundant), BSK_ALPHA_RNA_REDUNDANT, BSK_ALPHA_PROTEIN, -1 for "Unlimit".
 * bsk_batch_from_fastx = read_chunk + bsk_batch_from_ascii (alphabet < 0: use the guess). */
typedef struct bsk_fastx bsk_fastx;
int bsk_fastx_open(const char *path, bsk_fastx **out);
int bsk_fastx_read_chunk(bsk_fastx *f, uint64_t max_records, uint64_t max_bytes, uint64_t *n, const uint8_t **seq_bytes,
                         const uint64_t **seq_offsets, const uint8_t **name_bytes, const uint64_t **name_offsets,
                         const uint8_t **qual_bytes);
int bsk_fastx_info(const bsk_fastx *f, int *is_fastq, int *alphabet);
const char *bsk_fastx_error(const bsk_fastx *f);
void bsk_fastx_close(bsk_fastx *f);
int bsk_batch_from_fastx(bsk_ctx *ctx, bsk_fastx *f, uint64_t max_records, uint64_t max_bytes, int alphabet, bsk_batch **out,
                         uint64_t *n_records);

/* Block-parallel reader for PLAIN files (what ChunkChan's producer goroutine, reader.go:562-608, becomes when one thread cannot
 * feed a GPU): the same records in the same order as bsk_fastx_read_chunk, sequences only.  n_threads parser threads work
 * ahead on consecutive byte ranges of `piece_bytes` (0: 8 MiB); each guesses the first record start of its range, runs the
 * record state machine of the serial reader from there, and the consumer (bsk_fastx_par_next, one thread) validates every
 * guess against the previous piece's true end and re-parses a piece that started anywhere else -- so the records are the serial
 * reader's for ANY input (multi-line FASTQ merely runs serially); bsk_fastx_par_info reports how many pieces were re-parsed.
 * BGZF files (bgzip: gzip members of at most 64 KiB that record their own compressed size in a "BC" extra field) are taken too:
 * the members are located without inflating anything, pieces are ranges of the UNCOMPRESSED text, and every parser thread inflates
 * the blocks of its own range.  bsk_fastx_par_open: BSK_ERR_UNSUPPORTED for every other gzip file and for "-" (one serial stream:
 * use bsk_fastx_open), BSK_ERR_NOT_FASTX as above.
 * bsk_fastx_par_next: the next piece with at least one record, *piece == NULL at the end; an error inside a piece is returned
 * by the call after the one that delivered the records before it.  A piece stays valid until bsk_fastx_piece_release (any
 * thread), which must precede bsk_fastx_par_close. */
typedef struct bsk_fastx_par bsk_fastx_par;
typedef struct bsk_fastx_piece bsk_fastx_piece;
int bsk_fastx_par_open(const char *path, int n_threads, uint64_t piece_bytes, bsk_fastx_par **out);
int bsk_fastx_par_next(bsk_fastx_par *f, bsk_fastx_piece **piece);
int bsk_fastx_piece_data(const bsk_fastx_piece *piece, uint64_t *n, const uint8_t **seq_bytes, const uint64_t **seq_offsets);
void bsk_fastx_piece_release(bsk_fastx_par *f, bsk_fastx_piece *piece);
int bsk_fastx_par_info(const bsk_fastx_par *f, int *is_fastq, int *alphabet, uint64_t *reparsed_pieces);
const char *bsk_fastx_par_error(const bsk_fastx_par *f);
void bsk_fastx_par_close(bsk_fastx_par *f);

/* ---- compute -------------------------------------------------------------------
 * Runs the iterator/sketch named by p->kind over every read of the batch.
 * *result == NULL: a result is allocated; otherwise it is reused (bench loops).
 * Output layout (device, SoA, deterministic):
 *   refs[n] u64 = (first_tuple << 24) | n_tuples ; status[n] u8 ; hash[] u64 ; pos[] u32 (bit 31 strand)
 * read r owns tuples first_tuple + j * stride, j = 0 .. n_tuples - 1, in position order -- exactly
 * the sequence of (Next*() value, Index()) pairs the reference iterator yields.  stride is 1, or 64
 * when bit 63 of the reference word (BSK_REF_ROWS) is set: the minimizer / syncmer kernels for short
 * reads write a 64-read unit as ROWS (row j = the j-th tuple of each of the unit's 64 reads), which
 * lets whole rows leave the chip while the reads are still being hashed.
 * BSK_REF_FIRST / BSK_REF_COUNT / BSK_REF_STRIDE decode a reference word.
 * (The tuple arrays contain gaps; use bsk_result_fetch for a dense, rebased CSR copy on the host,
 * bsk_result_compact for one on the device.)
 * KMER / NTHASH / SIMHASH / PROT_HASH emit every position, so pos[] is implicit
 * (NULL): tuple j of read r is position j. */
int bsk_sketch(bsk_ctx *ctx, const bsk_batch *batch, const bsk_params *p, bsk_result **result);

/* Same, repeated: `warmup` untimed runs then `iters` runs, each bracketed by HIP
 * events on the context's stream.  kernel_ms[iters] (may be NULL) receives the
 * per-run duration of the sketch kernel; the call returns after the last run
 * completed.  With *result == NULL one untimed sizing run is done first; with an
 * existing result (from bsk_sketch on the same batch and params) only the
 * warmup + iters launches run.  Used by bench.py (roofline leg). */
int bsk_sketch_timed(bsk_ctx *ctx, const bsk_batch *batch, const bsk_params *p, bsk_result **result,
                     int warmup, int iters, float *kernel_ms);

/* Per-batch preparation a sketch with these parameters would do on its first call, done now (optional; bsk_sketch does it itself and
 * keeps it with the batch): the cut of a class plan (see bsk_result_class_plan), and -- for batches made with BSK_NO_BIN_EARLY=1 -- the
 * LENGTH-BINNED view of a ragged batch of short reads: the kernels walk the 64 reads of a unit in lock-step, so the reads of every chunk
 * of 4096 are grouped by length before units are formed (reference words and status bytes stay at the reads' own positions).  Since
 * round 5 that view is built with the batch (bsk_batch_from_ascii / _from_packed and the refills), in order of length, for whatever
 * parameters come.  *ms (may be NULL) receives the device time of the pass, 0 when the plan needs none.  The
 * reference has no counterpart: its cost is per base of one sequence at a time (sketches/sketch.go:46). */
int bsk_batch_prepare(bsk_ctx *ctx, const bsk_batch *batch, const bsk_params *p, float *ms);

int bsk_result_info(const bsk_result *r, uint64_t *n_reads, uint64_t *n_tuples, int *has_pos);
/* What ran: the name of the kernel the planner launched for this result, as a profiler shows it
 * ("k_minimizer_fast<11,32,true>", "k_syncmer<1>", "... (over tiles)"), its grid and the workgroups (= wavefronts) per CU
 * that grid amounts to.  The string lives as long as the result.  Any out-pointer may be NULL. */
int bsk_result_plan(const bsk_result *r, const char **kernel, int *grid, int *waves_per_cu);
/* Class plans.  The reference sketches one sequence at a time: a long contig costs its own bases, whatever else is in the file
 * (sketch.go:46, :85-94).  A batch whose lengths fall into several classes of the planner (150-base reads + a few of 400 or 5 000
 * bases) is therefore cut by length: the class with most bases runs over the whole batch with the other reads masked, every other class
 * as a batch of its own into the tail of the same result -- callers see one result, bsk_result_plan names every kernel
 * ("k_minimizer_pk<11,false> + k_minimizer_dense<11> [7 reads of 151..400 bases]").  *n_parts = classes besides the bulk (0: one plan),
 * *build_ms = device time of the passes that cut the batch (lists of the other classes' reads + the masked view), paid per bsk_sketch. */
int bsk_result_class_plan(const bsk_result *r, int *n_parts, float *build_ms);
/* Copy reads [first, first+count) to the host.  offsets[count+1] are rebased to 0;
 * hash/pos may be NULL; tuple_cap = capacity (in tuples) of hash[]/pos[]. */
int bsk_result_fetch(bsk_ctx *ctx, const bsk_result *r, uint64_t first, uint64_t count,
                     uint64_t *offsets, uint8_t *status, uint64_t *hash, uint32_t *pos,
                     uint64_t tuple_cap);
/* The status bytes of reads [first, first+count) alone. */
int bsk_result_fetch_status(bsk_ctx *ctx, const bsk_result *r, uint64_t first, uint64_t count, uint8_t *status);
/* The same with narrow side arrays, for streaming callers that move every tuple over the link (bsk_pipeline_*): offsets[count+1] as
 * u32 (scanned on the device: no round trip for the reference words), positions as u16 -- 15 bits + the strand in bit 15
 * (BSK_POS16_STRAND_BIT) -- i.e. 10 bytes per tuple + 5 per read instead of 12 + 17.  BSK_ERR_UNSUPPORTED when a position needs
 * more than 15 bits (reads of 32 768 bases or more) or the range holds 2^32 tuples; *n_tuples (may be NULL) receives the count. */
#define BSK_POS16_STRAND_BIT 0x8000u
#define BSK_POS16_MASK 0x7fffu
int bsk_result_fetch_narrow(bsk_ctx *ctx, const bsk_result *r, uint64_t first, uint64_t count, uint32_t *offsets, uint8_t *status,
                            uint64_t *hash, uint16_t *pos, uint64_t tuple_cap, uint64_t *n_tuples);
/* Dense device copy for device-side consumers (CSR: what bsk_result_fetch delivers to the host, left on the device): offsets[n+1]
 * (u64, offsets[n] = *n_tuples), hash[*n_tuples], pos[*n_tuples] (*pos = NULL for the kinds with implicit positions; pos may be NULL).
 * The arrays belong to the context and stay valid until the next bsk_result_compact on it (or bsk_ctx_destroy).  The reference has no
 * counterpart: its callers collect Next() values into a slice (e.g. sketches/sketch_test.go:93-104). */
int bsk_result_compact(bsk_ctx *ctx, const bsk_result *r, const uint64_t **offsets, const uint64_t **hash, const uint32_t **pos,
                       uint64_t *n_tuples);
/* Device pointers (valid until release / next bsk_sketch on this result); refs[] as described above.
 * Long sequences: when a DNA batch holds a sequence longer than the tile threshold (4096 bases, 512 for the every-position kinds; always from 2^24
 * bases on) the engine cuts the sequences into overlapping tiles, runs the same kernels over the tiles and stitches
 * the tile results back (exactly the tuples of the un-tiled iterator).  Such a result is
 * "wide": a sequence can own more than 2^24 tuples, so *refs is NULL and bsk_result_device_wide returns
 * first[n] / count[n] instead (u64 each).  bsk_result_fetch / bsk_result_digest work for both layouts.
 * circular = 1 tiles as well (the sequence with its first k-1 bases appended is one more long sequence), and so do syncmers
 * with s == k (every k-mer with its index, sketch.go:328-331: the w = 1 minimizer).
 * The two-strand k-mer mode (KMER with canonical = 0, iterator.go:713-723) tiles too: forward codes per tile, then the second
 * strand per sequence (the reverse complements of the forward codes, backwards).  Protein kinds on a DNA batch translate a
 * sequence of any length (up to 2^31 bases); the translation is then tiled like any long protein sequence. */
int bsk_result_device(const bsk_result *r, const uint64_t **refs, const uint8_t **status,
                      const uint64_t **hash, const uint32_t **pos);
int bsk_result_device_wide(const bsk_result *r, const uint64_t **first, const uint64_t **count);
/* Order-independent digest computed on device over the whole result:
 *   checksum = sum over tuples of hash * (2*position + 1)   (mod 2^64)
 *   status_counts[k] = number of reads whose status byte has bit pattern k set, for
 *   k in {SHORT, ILLEGAL, FIRST_WINDOW_TIE, HAS_NON_ACGT} (4 entries). */
int bsk_result_digest(bsk_ctx *ctx, const bsk_result *r, uint64_t *checksum, uint64_t *n_tuples,
                      uint64_t status_counts[4]);
/* Frees the result.  Its device arrays (a bench-size result reserves tens of GB) stay with the context for the next result of that context --
 * hipMalloc + hipFree of them cost a hundred kernel times -- up to eight buffers and 40 % of the device memory; the reserve goes back to the
 * device when any allocation of the library runs out of memory, and with bsk_ctx_destroy (BSK_NO_SPARE=1: freed at once). */
void bsk_result_release(bsk_result *r);

/* ---- streaming callers: bounded allocation and the end-to-end pipeline ------------------------
 * bsk_batch_refill_ascii: bsk_batch_from_ascii into an EXISTING batch object (*batch may be NULL the first time): the device
 * buffers are kept and only grow, so a caller that feeds chunk after chunk through one batch per stream allocates nothing in
 * steady state (hipFree synchronises the whole device and would serialise the streams).  The old contents are gone after the
 * call; on error *batch is NULL.  bsk_result_fetch likewise works out of grow-only buffers of its context.
 *
 * bsk_pipeline_fastx / bsk_pipeline_memory: the whole path host bytes -> tuples on the host with its stages overlapped --
 * one producer thread (the role of fastx's ChunkChan goroutine, seqio/fastx/reader.go:562-608) filling pinned chunks of
 * chunk_records records, n_streams workers, each with its own context, doing refill (H2D + pack) -> bsk_sketch ->
 * bsk_result_fetch into pinned memory (fetch_tuples != 0) or only a device digest (fetch_tuples == 0).  It uses nothing but
 * the calls above: a Go host gets the same overlap from one goroutine per stream.  stats->seconds is the wall time from the
 * first read to the last tuple; the per-stage seconds are summed over the workers (they overlap, so they add up to more
 * than `seconds`); checksum is the digest of bsk_result_digest summed over the chunks (positions are per sequence). */
int bsk_batch_refill_ascii(bsk_ctx *ctx, bsk_batch **batch, const uint8_t *bytes, const uint64_t *offsets, uint64_t n, int alphabet);
/* bsk_batch_refill_packed: the same for bsk_batch_from_packed (2-bit words + descriptors prepared on the host: a quarter of the bytes
 * over the link, no pack kernel).  bsk_pipeline_memory packs chunks of pure ACGT reads this way on its worker threads. */
int bsk_batch_refill_packed(bsk_ctx *ctx, bsk_batch **batch, const uint32_t *words, uint64_t n_words, const uint64_t *desc, uint64_t n);
typedef struct bsk_pipeline_stats {
    uint64_t records, bases, tuples, chunks, checksum;
    double seconds;              /* wall */
    double reader_seconds;       /* producer: inside the reader + the copy into pinned memory */
    double reader_wait_seconds;  /* producer: waiting for a free chunk buffer (the device side was the slower one) */
    double h2d_pack_seconds, kernel_seconds, fetch_seconds; /* summed over the workers */
    int32_t n_streams;
    int32_t reader_threads;      /* bsk_pipeline_fastx on a plain file: parser threads of the block-parallel reader (0: serial reader) */
    uint64_t reparsed_pieces;    /* ... and the pieces whose guessed record start was wrong (parsed again, serially) */
    double pin_seconds;          /* summed over all threads: getting the pinned chunk and result buffers (hipHostMalloc, or the pool of earlier runs) */
} bsk_pipeline_stats;
int bsk_pipeline_fastx(int device, const char *path, int alphabet /* -1: guess from the first record */, const bsk_params *p, int n_streams,
                       uint64_t chunk_records, int fetch_tuples, bsk_pipeline_stats *stats);
int bsk_pipeline_memory(int device, const uint8_t *bytes, const uint64_t *offsets, uint64_t n, int alphabet, const bsk_params *p,
                        int n_streams, uint64_t chunk_records, int repeat, int fetch_tuples, bsk_pipeline_stats *stats);
/* The same over SEVERAL GPUs of one node (devices[n_devices]; a device may be named more than once): one producer side, n_streams workers
 * per device, every worker takes the next chunk from the one queue -- the role of ChunkChan feeding W workers
 * (seqio/fastx/reader.go:562-608) with the workers spread over the node's GPUs.  Reads are independent: nothing crosses between devices,
 * and the statistics (records, tuples, the order-independent checksum) are the whole job's. */
int bsk_pipeline_fastx_multi(const int *devices, int n_devices, const char *path, int alphabet, const bsk_params *p, int n_streams,
                             uint64_t chunk_records, int fetch_tuples, bsk_pipeline_stats *stats);
int bsk_pipeline_memory_multi(const int *devices, int n_devices, const uint8_t *bytes, const uint64_t *offsets, uint64_t n, int alphabet,
                              const bsk_params *p, int n_streams, uint64_t chunk_records, int repeat, int fetch_tuples, bsk_pipeline_stats *stats);
/* Several files through ONE pipeline, n_readers of them (0: min(n_paths, 8)) read at the same time, each by its own producer thread --
 * the block-parallel reader for a plain file, the serial one for a gzip file (one zlib stream per file is how gzip input scales).
 * A file is closed as soon as its last chunk is on the device.  The statistics are those of the whole job. */
int bsk_pipeline_fastx_files(int device, const char *const *paths, int n_paths, int alphabet, const bsk_params *p, int n_streams, int n_readers,
                             uint64_t chunk_records, int fetch_tuples, bsk_pipeline_stats *stats);

/* ---- the pipeline with a consumer ----------------------------------------------------------------------------------------------
 * What fastx's ChunkChan is to the reference's callers (seqio/fastx/reader.go:562-608: every chunk of records, in input order, handed to
 * whoever ranges over the channel) with the sketching done on the way: open a pipeline over a source, then
 *     for (;;) { bsk_pipeline_next(pl, &c); if (!c) break;  ... use c ...;  bsk_pipeline_release(pl, c); }   bsk_pipeline_close(pl, &stats);
 * Chunks arrive in the order the producer queued them (= record order for one source); a chunk's arrays live in pinned host memory of the
 * pipeline until the chunk is released (the pool of output buffers is bounded: a consumer that holds 2 * workers + 2 chunks stalls the run).
 * sink selects what crosses the link for every chunk:
 *   BSK_SINK_TUPLES  every tuple, as bsk_result_fetch_narrow delivers it (offsets32 / pos16; offsets64 / pos32 when a chunk holds a read
 *                    of 32 768 bases or more, or 2^32 tuples) -- record i of the chunk owns hash[offsets[i] .. offsets[i+1]) in Next() order;
 *   BSK_SINK_SETS    the on-device reduction of bsk_result_sets: per record the ascending distinct hash values with hash <= MaxUint64 /
 *                    sets_scale (iterator.go:181-185; sets_scale <= 1: no filter) -- offsets32 + hash, n_values values;
 *   BSK_SINK_COUNTS  nothing but the counts and the device-side digest (bsk_result_digest) in `checksum`.
 * host_checksum != 0: the worker also folds what it fetched into `checksum` (tuples: the order-independent sum of bsk_result_digest; sets:
 * the sum of the values) -- the statistics-only entry points bsk_pipeline_fastx / _memory use this.
 * bsk_pipeline_next blocks until the next chunk is ready; *chunk == NULL with BSK_OK is the end of the input.  One consumer thread at a
 * time.  bsk_pipeline_close may be called at any point (it stops the run) and frees the pipeline; stats may be NULL. */
typedef struct bsk_pipeline bsk_pipeline;
enum { BSK_SINK_COUNTS = 0, BSK_SINK_TUPLES = 1, BSK_SINK_SETS = 2 };
typedef struct bsk_pipeline_config {
    const int *devices;      /* the GPUs of the run (a device may be named more than once) */
    int32_t n_devices;
    int32_t n_streams;       /* workers (context + HIP stream) per device */
    uint64_t chunk_records;  /* records per chunk (0: the reader's default) */
    int32_t sink;            /* BSK_SINK_* */
    int32_t sets_scale;      /* BSK_SINK_SETS: FracMinHash scale */
    int32_t alphabet;        /* bsk_alphabet, -1: guess from the first record (files) */
    int32_t host_checksum;
    int32_t n_readers;       /* several files: how many are read at the same time (0: min(n_paths, 8); chunk order = queue order) */
    int32_t reserved;
} bsk_pipeline_config;
typedef struct bsk_chunk {
    uint64_t sequence;       /* 0, 1, 2 ...: the delivery order */
    int32_t source_index;    /* which of the run's files */
    int32_t device;          /* the GPU that sketched it */
    uint64_t first_record;   /* index of the chunk's first record in its source */
    uint64_t n_records, n_bases, n_tuples;
    uint64_t n_values;       /* entries of hash[]: n_tuples (TUPLES), distinct filtered values (SETS), 0 (COUNTS) */
    uint64_t checksum;
    uint64_t link_bytes;     /* bytes this chunk's output moved device -> host */
    int32_t sink, has_pos;
    const uint32_t *offsets32; /* [n_records + 1], or NULL when offsets64 is set */
    const uint64_t *offsets64;
    const uint8_t *status;     /* [n_records] BSK_ST_* */
    const uint64_t *hash;
    const uint16_t *pos16;     /* TUPLES with positions: BSK_POS16_* encoding; NULL for implicit positions / SETS / when pos32 is set */
    const uint32_t *pos32;
    void *opaque;
} bsk_chunk;
int bsk_pipeline_open_fastx(const bsk_pipeline_config *cfg, const char *const *paths, int n_paths, const bsk_params *p, bsk_pipeline **out);
int bsk_pipeline_open_memory(const bsk_pipeline_config *cfg, const uint8_t *bytes, const uint64_t *offsets, uint64_t n, int repeat,
                             const bsk_params *p, bsk_pipeline **out);
int bsk_pipeline_next(bsk_pipeline *pl, const bsk_chunk **chunk);
int bsk_pipeline_release(bsk_pipeline *pl, const bsk_chunk *chunk);
int bsk_pipeline_close(bsk_pipeline *pl, bsk_pipeline_stats *stats);
const char *bsk_pipeline_error(const bsk_pipeline *pl);
/* Stops the run from ANY thread without freeing anything: a consumer blocked in bsk_pipeline_next (and every later call) returns -1, the
 * workers wind down.  What a host with a consumer thread of its own (Go: the goroutine behind Pipeline.Chunks(), the role of fastx's
 * ChunkChan goroutine, reader.go:562-608) calls BEFORE it joins that thread and closes: bsk_pipeline_close frees the pipeline and must
 * not overlap with a bsk_pipeline_next. */
int bsk_pipeline_cancel(bsk_pipeline *pl);
/* next / on_chunk / release until the end (or until on_chunk returns non-zero: BSK_ERR_STOPPED), then close: the callback form of the
 * loop above. */
typedef int (*bsk_chunk_fn)(void *user, const bsk_chunk *chunk);
int bsk_pipeline_run(bsk_pipeline *pl, bsk_chunk_fn on_chunk, void *user, bsk_pipeline_stats *stats);

/* The pipelines keep their pinned host buffers in a process-wide pool between calls (pinning is the start-up cost of a run:
 * pin_seconds); this returns the pooled memory to the system.  Safe at any time; buffers of a running pipeline are not affected. */
void bsk_pipeline_trim(void);

/* ---- multi-GPU: the one collective of the path ----------------------------
 * Reads shard by record; no tuple ever crosses GPUs.  What a job gathers at its end is a handful of u64 counters per GPU
 * (reads, bases, tuples, flagged reads ...): one all_gather over RCCL (xGMI inside a node), 8 * n_counters bytes per rank.
 * librccl is opened on first use; a single-GPU caller never loads it.
 *   one process (or thread) per GPU:  rank 0 calls bsk_comm_unique_id and hands the 128 bytes to the other ranks by whatever
 *     channel the host has (the job launcher's store, a Go channel); every rank then calls bsk_comm_init_rank on its own
 *     context and bsk_gather_counts after its work;
 *   one thread driving several contexts (a Go host with one goroutine per GPU that joins before reporting):
 *     bsk_comm_init_all once, bsk_gather_counts_all after the work -- mine[n][n_counters] in rank order, all[n][n_counters]
 *     as every rank received it (checked to be identical).
 * all[] is world * n_counters values in rank order.  The calls block until the data is on the host. */
#define BSK_UNIQUE_ID_BYTES 128
#define BSK_MAX_COUNTERS 16
int bsk_comm_unique_id(uint8_t *id /* [BSK_UNIQUE_ID_BYTES] */);
int bsk_comm_init_rank(bsk_ctx *ctx, const uint8_t *id, int rank, int world);
int bsk_comm_init_all(bsk_ctx *const *ctxs, int n);
int bsk_gather_counts(bsk_ctx *ctx, const uint64_t *mine, int n_counters, uint64_t *all);
int bsk_gather_counts_all(bsk_ctx *const *ctxs, int n, const uint64_t *mine, int n_counters, uint64_t *all);
void bsk_comm_destroy(bsk_ctx *ctx);

/* ---- sketch sets ----------------------------------------------------
 * The distinct hash values of a result in ascending order -- what the reference's consumers (kmcp, unikmer) build from
 * the Next() stream of a sketch and keep on disk: collect, sort, de-duplicate.  scope: one set per sequence or one set
 * for the whole batch.  scale > 1: FracMinHash, keep hash <= MaxUint64/scale (the rule of iterator.go:181-185).
 * Computed on the device (rocPRIM segmented radix sort + scan); offsets[n_sets+1] and values[] stay there until
 * bsk_sets_release.  At most 2^32 values per call. */
typedef struct bsk_sets bsk_sets;
enum { BSK_SETS_PER_SEQUENCE = 0, BSK_SETS_WHOLE_BATCH = 1 };
int bsk_result_sets(bsk_ctx *ctx, const bsk_result *r, int scope, int scale, bsk_sets **out);
/* The same into an existing object (*sets may be NULL the first time): the device arrays are kept and only grow -- a streaming caller
 * allocates nothing in steady state.  On error the object is released and *sets is NULL. */
int bsk_result_sets_reuse(bsk_ctx *ctx, const bsk_result *r, int scope, int scale, bsk_sets **sets);
/* All sets to the host in narrow form: u32 offsets[n_sets + 1] (narrowed on the device) + the values, copied on the context's stream
 * (4 bytes per set + 8 per value over the link). */
int bsk_sets_fetch_narrow(bsk_ctx *ctx, const bsk_sets *s, uint32_t *offsets, uint64_t *values, uint64_t value_cap);
int bsk_sets_info(const bsk_sets *s, uint64_t *n_sets, uint64_t *n_values);
int bsk_sets_fetch(bsk_ctx *ctx, const bsk_sets *s, uint64_t first, uint64_t count, uint64_t *offsets, uint64_t *values,
                   uint64_t value_cap);
int bsk_sets_device(const bsk_sets *s, const uint64_t **offsets, const uint64_t **values);
void bsk_sets_release(bsk_sets *s);

#ifdef __cplusplus
}
#endif
#endif /* BIOSKETCH_H */
