// host_types.hpp -- host-side objects behind the opaque handles of include/biosketch.h, shared by the translation units
// of libbiosketch.so (biosketch.hip: batches, dispatch; sets.hip: sorted unique hash sets).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>
#include <vector>

#include "biosketch.h"
#include "device_common.hpp"

// ------------------------------------------------------------------------------------
// Developer switches (DESIGN.md "Switches"): read from the environment ONCE, when the context is created, or again on
// bsk_ctx_reload_options (the test suite flips them inside one process) -- never on the bsk_sketch path.
struct BskOpts {
    bool force_generic = false, no_mixed = false, no_dense = false, no_pk = false, no_ring = false, no_pkd = false, no_side_early = false, no_side_dense = false, ring = false, no_bin = false, no_bin_early = false, compact = false, no_spare = false, no_tiles = false, no_tile_cache = false, no_tile_defer = false, tile_dense = false, no_group_gather = false, timing = false,
         no_fused_translate = false, sets_no_small = false;
    int syn_margin = 2;
    u32 test_overflow = 0;   // BSK_TEST_OVERFLOW (tests): pretend an overflow flag once per call -- 1: in a timed re-run (BSK_RESIZE), 2: a class plan's part while sizing, 4: ... in a timed re-run (BSK_REPLAN_CLASS)
    bool no_syn_long = false;
    u32 pf_density = 0;      // BSK_PF_DENSITY (tests): the expected selections per read up to which k_syncmer_pf / _pfl are planned (0: 86 % of what a unit's emit list holds per read -- 1 024 / 1 792 tuples per 64 reads; beyond the list the unit's last reads go to the exact machine)
    bool no_syn_pf = false;  // BSK_NO_SYN_PF: syncmers on k_syncmer_pk / _pkl also where the fused-emit kernel (k_syncmer_pf, round 6) is planned
    bool syn_sel = false;    // BSK_SYN_SEL (make EXPERIMENTS=1): the two-pass syncmer plan, measured and not planned (kernels_syncmer_sel.hpp)
    bool no_class = false;   // BSK_NO_CLASS: one plan per batch, keyed on the longest read (rounds 1-4)
    u32 class_min = 16384;   // BSK_CLASS_MIN: batches below this many reads keep one plan
    bool class_view = false;   // BSK_CLASS_VIEW: class plans always cut on the device (k_class_cut + a view of the batch), also where the host's list would do
    bool class_force = false;  // BSK_CLASS_FORCE: cut wherever the planner's choice changes, whatever the cost model says (tests: small batches)
    u32 wpr = 0, seg = 0, dense_min = 21 /* PlannerTable::dense_min */, ring_max = 0, ring_sel10 = 0 /* dev: ring_rows' selections per window x (w + 1), in tenths (0: PlannerTable::slab_sel_num) */, bin_min = 1024, waves_per_cu = 0, tile_min = 0, tile_pos = 0;  // tile_min 0: the kind's default
    void load();  // biosketch.hip
};

struct bsk_ctx {
    BskOpts opt;
    int device = 0;
    int cus = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // per-launch synchronisation scratch
    u32 *d_ticket = nullptr;    // [32]: 0..7 a launch's tickets and overflow flags, 8..15 class cuts' cursors, 16 the overflow flags of a class plan's PARTS (their own launches on the side context reset 0..7 one after another: k_fold_flags)
    u64 *d_total = nullptr;     // [1]
    u64 *d_lookback = nullptr;  // [lookback_cap]
    size_t lookback_cap = 0;
    u64 *d_ring_h = nullptr;  // runtime-w ring scratch
    u32 *d_ring_p = nullptr;
    size_t ring_cap = 0;  // entries
    u64 *h_pinned = nullptr;  // [8] pinned host words for small read-backs
    u8 *d_lut = nullptr;      // codon tables of `lut_table` (kernels_translate.hpp layout)
    int lut_table = 0;
    // tiled calls: grow-only temporaries (hipMalloc / hipFree of ten buffers per call cost more than the kernels)
    void *tmp[32] = {};      // 0-9: tiled calls / sets, 12-14: bsk_result_fetch, 16-19: bsk_result_compact, 20: bsk_sets_fetch_narrow, 21-23: class plans, 24-27: two-pass syncmers, 28-29: the ASCII side launch's own reference words / status bytes
    size_t tmp_cap[32] = {};
    u64 *h_refs = nullptr;   // pinned staging of bsk_result_fetch (refs down, offsets up), grow-only
    size_t h_refs_cap = 0;
    struct bsk_result *tile_res = nullptr;  // tile-level result of the previous tiled call, reused
    // Result arrays a released result leaves behind (biosketch.hip: spare_give / spare_take).  hipFree + hipMalloc of the arrays of a bench-size
    // result (10^8 reads: 32 GB) cost 1.2-3 s a call -- a hundred kernel times; a caller that releases every result and sketches the next
    // batch gets the last ones back instead.  Bounded (SPARE_SLOTS buffers, 40 % of the device memory), flushed when any allocation of the
    // library runs out of memory (host_internal.hpp: hipMalloc retries once after the flush) and with the context.
    static constexpr int SPARE_SLOTS = 8;
    struct Spare {
        void *p = nullptr;
        size_t bytes = 0;
    } spare[SPARE_SLOTS];
    size_t spare_bytes = 0, spare_limit = 0;
    // the one collective of the path (comm.cpp): an RCCL communicator over the GPUs that share a job
    void *comm = nullptr;  // ncclComm_t
    int comm_rank = 0, comm_world = 0;
    u64 *d_comm = nullptr;  // [(world + 1) * BSK_MAX_COUNTERS] device staging of bsk_gather_counts
    // class plans: the parts (the classes besides the bulk) run on a SIDE context -- a stream and scratch of their own -- so that their
    // small, latency-bound launches overlap with the bulk's kernel instead of queueing in front of it; two events order the two streams
    bsk_ctx *side = nullptr;              // created on first use, destroyed with this context
    hipEvent_t ev_side_done = nullptr, ev_adopted = nullptr, ev_mix0 = nullptr, ev_mix1 = nullptr, ev_tiled = nullptr;
    bool adopted_recorded = false;
    struct ClassSet *cls = nullptr;       // the class plan a run_planned / launch in progress belongs to (biosketch.hip: run_classed)
    bool no_side_fast = false;            // run_planned: a staged side kernel's region overflowed, plan the general one
    struct bsk_result *cls_owner = nullptr;  // the result whose class plan the pooled lists / views (tmp 21-23) currently describe
    u64 sel_need = 0;           // two-pass syncmers: the dense region a call that is being sized again needs (run_planned)
    bool no_syn_pk = false;     // set while a call falls back from k_syncmer_pk / k_minimizer_pk (the list of reads for the exact machine filled up)
    bool no_prot_fast = false;
    bool defer = false;         // run_planned: plan, size from the plan's own bounds, launch ONCE and read nothing back (sketch_tiled's path without host round trips: the caller looks at the flags behind its own last synchronisation)
    bool tile_sync = false;     // sketch_tiled: the round-trip path (tile count, sizing run and totals read on the host) -- set while a call falls back after the deferred path's flags showed an overflow
    bool tile_async = false;    // sketch_tiled on a class plan's side context: not even the last synchronisation -- totals stay on the device, the flags in d_ticket[24] (launch_parts folds them into the parent's)
    bool tile_was_async = false;  // ... what the last sketch_tiled on this context did
    bool in_resize = false;     // a timed call that outgrew its sized regions is being sized again (once per call: run_planned_resizing)
    int cls_round = 0;          // run_classed: which sizing round of the class plan is running (tests: BSK_TEST_OVERFLOW fires in round 0 only)
    bool part_grow = false;     // ... and a class plan's parts get twice their previous overflow regions (run_classed)
    bool no_dense = false;      // same for the dense-minimizer kernel (per-read slabs)  // set while a call falls back from the per-sequence-slab protein kernel
};

// Exact histogram of a batch's sequence lengths, built on the host where the batch is created (the lengths pass through the host there
// anyway): buckets of 16 bases below 1024, quarter octaves above.  What the per-length-class plans of biosketch.hip ("class plans") are
// cut from: which lengths occur, how many reads and bases each class holds -- without a device pass or a read-back.
struct LenHist {
    static constexpr int NB = 160;
    u64 cnt[NB] = {}, bases[NB] = {};
    u32 hi[NB] = {};  // the longest sequence of the bucket
    u32 lo[NB];       // ... and the shortest (valid where cnt != 0)
    LenHist() {
        for (int i = 0; i < NB; ++i) lo[i] = 0xffffffffu;
    }
    static int bucket(u64 L) {
        if (L < 1024) return (int)(L >> 4);
        const int lg = 63 - __builtin_clzll(L);  // >= 10
        return 64 + 4 * (lg - 10) + (int)((L >> (lg - 2)) & 3);
    }
    void add(u64 L) {
        const int i = bucket(L);
        cnt[i]++;
        bases[i] += L;
        if ((u32)L > hi[i]) hi[i] = (u32)L;
        if ((u32)L < lo[i]) lo[i] = (u32)L;
    }
};

struct bsk_batch {
    bsk_ctx *ctx = nullptr;
    int alphabet = BSK_ALPHA_DNA;
    int pairs = BSK_ALPHA_DNA;  // nucleotide batches: the alphabet whose PairLetter the two-strand k-mer mode applies (BSK_ALPHA_DNA = DNAredundant, 2..5)
    u64 n = 0, n_bases = 0, n_words = 0, n_nonacgt = 0;
    u32 maxlen = 0;
    u32 side_maxlen = 0;   // class views: the longest read of the WHOLE batch (the ASCII side launch covers every flagged read, whatever its class); 0: maxlen
    u32 uniform_len = 0;  // != 0: every read has this length (synthetic batches)
    u32 *words = nullptr;
    u64 *desc = nullptr;   // NULL when a sequence has 2^24 bases or more: fw + llen then locate the sequences (tiled runs only)
    u64 *fw = nullptr;     // [n] first word
    u64 *llen = nullptr;   // [n] bases
    u64 *adesc = nullptr;  // tile batches over ASCII: (first_byte << 24) | n_bases per tile
    bool alias = false;    // words / ascii belong to another batch (tile batches)
    bool borrowed = false; // a VIEW of another batch (class plans): nothing but desc / bdesc / bflags / adesc is its own
    LenHist *hist = nullptr;  // DNA batches of varying length created from host data; NULL: unknown (synthetic, tiles, translated)
    // ... and, when at most a twentieth of the sequences lie outside the histogram's fullest bucket, those sequences: (index << 32) | bases,
    // ascending.  A class plan whose bulk holds that bucket then needs NO device pass to cut the batch: the other classes' lists are picked
    // from here on the host and the bulk's kernel masks by length itself (KArgs::cls_*).
    std::vector<u64> *odd = nullptr;
    u64 *d_odd = nullptr;  // the same list on the device (lists of 65 536 and more: class plans split them there)
    int modal_bucket = -1;
    u32 *wbits = nullptr;  // one bit per packed word: the word holds a non-ACGT letter (batches that may be tiled)
    u32 *subset = nullptr; // reads with a non-ACGT letter, ascending (side launch of the ASCII kernels); nsub = n_nonacgt
    u64 nsub = 0;
    u8 *rflags = nullptr;
    // length-binned view of desc / rflags (ensure_binned, planner.hip): built on the first sketch call that plans it, kept with the batch
    mutable u64 *bdesc = nullptr;
    mutable u8 *bflags = nullptr;
    mutable u32 bin_gran = 0, bin_lo = 0;  // bases per length class (above bin_lo) the view was built with (0: not built)
    mutable bool bin_early = false;        // the view was built with the batch (bin_with_batch, biosketch.hip): classes finer than any plan's, no pass per plan
    mutable size_t c_bdesc = 0, c_bflags = 0;
    u8 *ascii = nullptr;  // DNA: kept only when some read has a non-ACGT byte; protein: always
    u64 *aoff = nullptr;
    u64 device_bytes = 0;
    // capacities (bytes) of the device buffers, and the ASCII buffers parked while a batch is pure ACGT: a streaming caller
    // re-fills one batch object per stream (bsk_batch_refill_ascii) instead of allocating per chunk -- hipFree synchronises the
    // whole device and would serialise the streams
    size_t c_words = 0, c_desc = 0, c_rflags = 0, c_ascii = 0, c_aoff = 0;
    u8 *spare_ascii = nullptr;
    u64 *spare_aoff = nullptr;
};

struct ClassSet;  // biosketch.hip: the parts of a class plan (sub-batches, their results, the lists of their reads)
struct bsk_result {
    bsk_ctx *ctx = nullptr;
    u64 n = 0, cap = 0, n_tuples = 0;
    u64 alloc_cap = 0;  // tuples hash[] / pos[] were allocated for: cap + tail_cap
    u64 tail_cap = 0;   // class plans: tuples behind `cap` that hold the adopted parts' slabs
    bool arrays_borrowed = false;  // hash / pos point into another result's tail (a part of a class plan)
    ClassSet *classes = nullptr;   // class plan of the last sizing call (owned)
    u64 n_cap = 0;  // reads the refs / status arrays were allocated for
    u64 ovf_cap = 0;  // slab kernels: tuples reserved (inside cap) for units that outgrow their slab
    u64 main_cap = 0; // tuples [0, main_cap) belong to the main launch, [main_cap, cap) to the side launch (mixed batches)
    int kind = 0, has_pos = 0;
    u64 *refs = nullptr;  // per read: (first_tuple << 24) | n_tuples ; NULL for wide results
    u64 *wfirst = nullptr, *wcount = nullptr;  // wide results (tiled long sequences): first tuple and tuple count per sequence
    u8 *status = nullptr;
    u64 *hash = nullptr;
    u32 *pos = nullptr;
    char plan[320] = "";   // what ran: kernel name of the last launch into this result (bsk_result_plan)
    int plan_grid = 0, plan_per_cu = 0;
    // The plan the arrays were SIZED for (run_planned, launch.hip): bsk_sketch_timed on an existing result repeats exactly this plan --
    // a fresh plan could want more rows per unit than `cap` holds (k_minimizer_ring after a fall-back to 32-row slabs) and write past it.
    bool plan_valid = false;
    bool unit_rows = false;  // the last launch wrote unit rows (BSK_REF_ROWS: k_minimizer_ring): what the group gathers of sets.hip dispatch on
    u64 plan_n = 0, plan_bases = 0;
    u32 plan_maxlen = 0;
    int plan_circ = 0;
    bsk_params plan_params = {};
    alignas(8) unsigned char plan_blob[192] = {};
};

static inline int fail_hip(bsk_ctx *ctx, hipError_t e, const char *what) {
    if (ctx) {
        char buf[512];
        snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
        ctx->err = buf;
    }
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? BSK_ERR_NOMEM : BSK_ERR_DEVICE;
}
static inline int fail_arg(bsk_ctx *ctx, const char *what) {
    if (ctx) ctx->err = what;
    return BSK_ERR_ARG;
}
#define HIPCHK(ctx, call)                                        \
    do {                                                         \
        hipError_t e__ = (call);                                 \
        if (e__ != hipSuccess) return fail_hip(ctx, e__, #call); \
    } while (0)

bool spare_flush_all() __attribute__((visibility("hidden")));  // biosketch.hip
// Every device allocation of the library's host code (this header is in all of its translation units): when the device is out of memory, what the contexts keep in reserve (released
// results' arrays) goes back first, then the allocation is tried once more.
static inline hipError_t bsk_malloc_retry(void **p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory && spare_flush_all()) {
        (void)hipGetLastError();
        e = hipMalloc(p, bytes);
    }
    return e;
}
#define hipMalloc(ptr, bytes) bsk_malloc_retry((void **)(ptr), (bytes))
