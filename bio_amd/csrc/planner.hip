// planner.hip -- which kernel runs a (batch, params) pair, on how many workgroups, and how the tuple arrays are organised
// (make_plan / make_plan_enc; the thresholds are planner_table.hpp's), the binned view, scratch, capacity estimates, kernel names.
#include "host_internal.hpp"
#include "kernels_host.hpp"

// ------------------------------------------------------------------------------------
// dispatch
// ------------------------------------------------------------------------------------
int validate(const bsk_params *p, int alphabet) {
    switch (p->kind) {
        case BSK_NTHASH:  // NewHashIterator iterator.go:616
            if (p->k < 1) return BSK_ERR_INVALID_K;
            break;
        case BSK_KMER:  // NewKmerIterator iterator.go:669 ; kmers.Encode rejects k > 32 at the first NextKmer
            if (p->k < 1) return BSK_ERR_INVALID_K;
            if (p->k > 32) return BSK_ERR_K_TOO_LARGE;
            break;
        case BSK_SIMHASH:  // NewSimHashIterator iterator.go:114-126
            if (p->k < 1) return BSK_ERR_INVALID_K;
            if (p->k >= 65535) return BSK_ERR_K_TOO_LARGE;
            if (p->m < 4 || p->m > p->k) return BSK_ERR_INVALID_M;
            if (p->scale < 1 || p->scale > p->k - p->m + 1) return BSK_ERR_INVALID_SCALE;
            if (p->k - p->m + 1 > 32767) return BSK_ERR_UNSUPPORTED;  // the reference's int16 counters would wrap
            break;
        case BSK_MINIMIZER:  // NewMinimizerSketch sketch.go:86-91
            if (p->k < 1) return BSK_ERR_INVALID_K;
            if (p->w < 1) return BSK_ERR_INVALID_W;
            break;
        case BSK_SYNCMER:  // NewSyncmerSketch sketch.go:143-148
            if (p->k < 1) return BSK_ERR_INVALID_K;
            if (p->s > p->k || p->s <= 0) return BSK_ERR_INVALID_S;
            break;
        case BSK_PROT_HASH:  // NewProteinIterator iterator-protein.go:47
            if (p->k < 1) return BSK_ERR_INVALID_K;
            break;
        case BSK_PROT_MINIMIZER:  // NewProteinMinimizerSketch sketch-protein.go:63-72
            if (p->k < 1) return BSK_ERR_INVALID_K;
            if (p->w < 1) return BSK_ERR_INVALID_W;
            break;
        default: return BSK_ERR_ARG;
    }
    const bool prot = p->kind == BSK_PROT_HASH || p->kind == BSK_PROT_MINIMIZER;
    if (!prot && alphabet == BSK_ALPHA_PROTEIN) return BSK_ERR_UNSUPPORTED;  // nucleotide sketches of a protein batch
    return BSK_OK;  // protein kinds on a DNA batch: translated first (sketch_impl)
}

int ensure_scratch(bsk_ctx *ctx, size_t nunits, size_t ring_entries) {
    if (ctx->lookback_cap < nunits) {
        (void)hipFree(ctx->d_lookback);
        ctx->d_lookback = nullptr;
        ctx->lookback_cap = 0;
        HIPCHK(ctx, hipMalloc(&ctx->d_lookback, nunits * sizeof(u64)));
        ctx->lookback_cap = nunits;
    }
    if (ctx->ring_cap < ring_entries) {
        (void)hipFree(ctx->d_ring_h);
        (void)hipFree(ctx->d_ring_p);
        ctx->d_ring_h = nullptr;
        ctx->d_ring_p = nullptr;
        ctx->ring_cap = 0;
        HIPCHK(ctx, hipMalloc(&ctx->d_ring_h, ring_entries * sizeof(u64)));
        HIPCHK(ctx, hipMalloc(&ctx->d_ring_p, ring_entries * sizeof(u32)));
        ctx->ring_cap = ring_entries;
    }
    return BSK_OK;
}

// list of the reads that carry a non-ACGT letter (they are few in real data): the fast 2-bit kernels then run over the
// whole batch and the general ASCII kernels re-do only these reads in a side launch (make_plan: "mixed")
int build_subset(bsk_ctx *ctx, bsk_batch *b) {
    (void)hipFree(b->subset);
    b->subset = nullptr;
    b->nsub = 0;
    if (!b->n || !b->n_nonacgt || !b->rflags || b->n >= (1ULL << 32)) return BSK_OK;
    const u32 nunits = (u32)((b->n + 63) / 64);
    int rc = ensure_scratch(ctx, lb_words_with_heads(nunits), 0);
    if (rc != BSK_OK) return rc;
    HIPCHK(ctx, hipMalloc(&b->subset, b->n_nonacgt * sizeof(u32)));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_ticket, 0, 4 * sizeof(u32), ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_lookback, 0, lb_words_with_heads(nunits) * sizeof(u64), ctx->stream));
    hipLaunchKernelGGL(k_compact_flags, dim3(std::min<u32>(nunits, (u32)ctx->cus * 8)), dim3(64), 0, ctx->stream, b->rflags, b->n, nunits,
                       ctx->d_ticket, ctx->d_lookback, b->subset);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    b->nsub = b->n_nonacgt;
    return BSK_OK;
}

void plan_record(bsk_result *res, const bsk_batch *b, const bsk_params *p, int circ_ext, const Plan &pl) {
    memcpy(res->plan_blob, &pl, sizeof pl);
    res->plan_params = *p;
    res->plan_n = b->n;
    res->plan_bases = b->n_bases;
    res->plan_maxlen = b->maxlen;
    res->plan_circ = circ_ext;
    res->plan_valid = true;
}
bool plan_recall(const bsk_result *res, const bsk_batch *b, const bsk_params *p, int circ_ext, Plan &pl) {
    if (!res->plan_valid || res->plan_n != b->n || res->plan_bases != b->n_bases || res->plan_maxlen != b->maxlen || res->plan_circ != circ_ext ||
        memcmp(&res->plan_params, p, sizeof *p) != 0)
        return false;
    memcpy(&pl, res->plan_blob, sizeof pl);
    return true;
}

// per-read slabs are sized by the LONGEST read: acceptable only while that does not blow the result arrays up (a batch of
// short reads with one long outlier would otherwise reserve the outlier's slab for every read)
bool slab_budget_ok(const bsk_batch *b, u64 slab_read) {
    const double mean = b->n ? (double)b->n_bases / (double)b->n : 0.0;
    return (double)b->maxlen <= 4.0 * mean + 64.0 || (double)b->n * (double)slab_read * 12.0 < 256.0 * 1024 * 1024;
}

// Length binning pays when the reads of a unit end more than half a block of `step` k-mers apart (KArgs::binned, k_bin_desc): ragged
// batches of short reads on the lock-step kernels.  Returns the bases per length class (a multiple of step, at most 63 classes), 0: no.
u32 bin_gran_for(const bsk_ctx *ctx, const bsk_batch *b, int step) {
    if (ctx->opt.no_bin || b->uniform_len || !b->desc || b->alias || b->maxlen >= 4096u || b->n < (u64)ctx->opt.bin_min || step < 1) return 0;
    const double mean = (double)b->n_bases / (double)b->n;
    if (((double)b->maxlen - mean) * 2.0 < (double)step) return 0;
    u32 g = (u32)step;
    while (b->maxlen / g > 61u) g += (u32)step;
    return g;
}
// the batch's length-binned descriptors for classes of `gran` bases above `lo`, built on the context's stream on first use and kept
// with the batch
int ensure_binned(bsk_ctx *ctx, const bsk_batch *b, u32 lo, u32 gran, u32 mlo, u32 mhi, u32 mpretend, bool fine) {
    if (b->bin_early && !mhi && b->bdesc) return BSK_OK;  // built with the batch, finer than any plan's classes (bin_with_batch)
    if (b->bin_gran == gran && b->bin_lo == lo && b->bdesc && !b->bin_early) return BSK_OK;
    b->bin_early = false;
    const size_t need_d = (size_t)b->n * sizeof(u64), need_f = b->rflags ? (size_t)b->n : 0;
    if (b->c_bdesc < need_d) {
        (void)hipFree(b->bdesc);
        b->bdesc = nullptr;
        b->c_bdesc = 0;
        HIPCHK(ctx, hipMalloc(&b->bdesc, need_d + need_d / 8));
        b->c_bdesc = need_d + need_d / 8;
    }
    if (b->c_bflags < need_f) {
        (void)hipFree(b->bflags);
        b->bflags = nullptr;
        b->c_bflags = 0;
        HIPCHK(ctx, hipMalloc(&b->bflags, need_f + need_f / 8));
        b->c_bflags = need_f + need_f / 8;
    }
    const u64 nchunks = (b->n + 4095) / 4096;
    if (fine)
        hipLaunchKernelGGL(k_bin_desc<128>, dim3((unsigned)std::min<u64>(nchunks, (u64)ctx->cus * 4)), dim3(512), 0, ctx->stream, b->desc, b->rflags, b->n, lo,
                           gran, b->bdesc, b->rflags ? b->bflags : nullptr, mlo, mhi, mpretend);
    else
        hipLaunchKernelGGL(k_bin_desc<64>, dim3((unsigned)std::min<u64>(nchunks, (u64)ctx->cus * 4)), dim3(512), 0, ctx->stream, b->desc, b->rflags, b->n, lo,
                           gran, b->bdesc, b->rflags ? b->bflags : nullptr, mlo, mhi, mpretend);
    HIPCHK(ctx, hipGetLastError());
    b->bin_gran = gran;
    b->bin_lo = lo;
    return BSK_OK;
}

// rows of 64 tuples in a unit's slab (kernels_ring.hpp): a read selects 2 / (w + 1) of its windows; + 30 % + 6, in whole groups of four
// rows (150 bp, w = 11: 32 rows).  A read with more goes to the exact machine's list.
static u64 ring_rows(double nwin, int w, double sel = PlannerTable::slab_sel_num) {
    const double nw = std::max(nwin, 1.0);
    return ((u64)std::min(nw, std::ceil(nw * sel / (w + 1.0)) + 6.0) + 3) & ~(u64)3;
}


static bool which_is_fast(Which w) {
    return w == K_MIN_FAST || w == K_NT_FAST || w == K_SYN_FAST || w == K_SIM_FAST || w == K_MIN_DENSE || w == K_MIN_SEG || w == K_MIN_WPR || w == K_MIN_PK || w == K_SYN_PK || w == K_MIN_RING || w == K_SYN_SEL || w == K_MIN_PKD;
}


int make_plan(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p, Plan &pl) {
    const bool has_n = b->alphabet == BSK_ALPHA_DNA && b->n_nonacgt > 0;
    // reads with a non-ACGT letter (up to 90 %: the side launch costs flagged/60 against 1/690 Gbases/s for the fast kernel, so this wins
    // almost always): plan the batch as 2-bit; if that lands on a fast kernel, the flagged reads are
    // re-done by the general ASCII kernel in a side launch.  Otherwise the whole batch runs on the ASCII kernels.
    // (round 5: with the flagged reads on a STAGED ASCII kernel -- K_MIN_DENSE_A / K_SYN_FAST_A below -- the pair wins at any share: a batch with
    // an IUPAC letter in every read ran on the general ASCII kernel at 63 Gbases/s, scripts/dev/scan_plans.py)
    const bool few_flagged = has_n && b->nsub * 10 <= b->n * 9;
    if (has_n && b->subset && !ctx->opt.no_mixed) {
        Plan t;
        int rc = make_plan_enc(ctx, b, p, t, false);
        if (rc != BSK_OK) return rc;
        if (which_is_fast(t.which)) {
            Plan sd;
            bsk_batch sb = *b;  // shallow view with the side launch's shape
            sb.n = b->nsub;
            rc = make_plan_enc(ctx, &sb, p, sd, true);
            if (rc != BSK_OK) return rc;
            pl = t;
            pl.mixed = true;
            pl.side_which = sd.which;
            pl.side_nunits = sd.nunits;
            pl.side_ring_w = sd.ring_w;
            pl.side_grid = sd.grid;
            pl.ring_entries = std::max(pl.ring_entries, sd.ring_entries);
            // minimizers: the flagged reads on k_minimizer_dense<W, false, true> -- the staged 64-bit machine fed from ASCII, one slab of a
            // tuple per window for every read (nothing to outgrow) -- while those slabs stay below 4 GB (round 5: the general kernel, whose
            // window lives in global memory, ran 1 % of the reads in a third of the call: 1.5 10^9 bases of 150-base reads 755 against
            // 1 150 Gbases/s, 1 000-base reads 355 against 600)
            if (p->kind == BSK_MINIMIZER && sd.which == K_MIN_GEN_A && dense_minimizer_supported(p->w) && !p->circular && !b->adesc && b->aoff && std::max(b->maxlen, b->side_maxlen) < 32768u &&
                !ctx->opt.force_generic && !ctx->opt.no_side_dense && !ctx->no_side_fast) {
                const u32 longest = b->side_maxlen ? b->side_maxlen : b->maxlen;  // (a class view's side launch takes the other classes' flagged reads too)
                const u64 nwin_max = longest + 2 > (u32)(p->k + p->w) ? (u64)longest - p->k - p->w + 2 : 1;
                const u64 slab = (nwin_max + 15) & ~(u64)15;
                const u64 units = (b->nsub + 63) / 64;
                u64 budget = 4ULL << 30;
                if (units * 64 * slab * 12 > budget) {  // (many flagged reads: up to 24 GB of side region where the device has four times that free)
                    size_t fr = 0, tot = 0;
                    if (hipMemGetInfo(&fr, &tot) == hipSuccess && (u64)fr >= (96ULL << 30)) budget = 24ULL << 30;
                    else (void)hipGetLastError();
                }
                if (units * 64 * slab * 12 <= budget) {
                    pl.side_which = K_MIN_DENSE_A;
                    pl.side_slab = slab;
                    pl.side_nunits = (u32)units;
                    pl.side_grid = (int)std::max<u64>(1, std::min<u64>(units, (u64)ctx->cus * (u64)dense_minimizer_ascii_blocks_per_cu(p->w)));  // (a ticket is one unit)
                    pl.side_ring_w = 0;
                }
            }
            // syncmers likewise: k_syncmer_fast<W, false, true> (unit slabs of 28 tuples per read + an overflow region for the units with a
            // read beyond them, all inside the side region; 1 % of 150-base reads flagged: 603-621 against 827 Gbases/s clean)
            if (p->kind == BSK_SYNCMER && sd.which == K_SYN_A && fast_syncmer_supported(p->k, p->s) && p->k - p->s <= 24 /* (k_syncmer_ascii.hip's list) */ && p->s != p->k && !p->circular && !b->adesc && b->aoff &&
                std::max(b->maxlen, b->side_maxlen) < 32768u && !ctx->opt.force_generic && !ctx->opt.no_side_dense && !ctx->no_side_fast) {
                const u64 units = (b->nsub + 63) / 64;
                pl.side_which = K_SYN_FAST_A;
                pl.side_slab = BSK_SYN_CAP;
                pl.side_nunits = (u32)units;
                pl.side_grid = (int)std::max<u64>(1, std::min<u64>(units, (u64)ctx->cus * 4));  // (a ticket is one unit)
                pl.side_ring_w = 0;
            }
            if (few_flagged || pl.side_which == K_MIN_DENSE_A || pl.side_which == K_SYN_FAST_A) return BSK_OK;
            pl = Plan();  // nearly every read flagged and only the general ASCII kernel to take them: one ASCII plan for the batch
        }
    }
    return make_plan_enc(ctx, b, p, pl, has_n);
}

// k_nthash_fast<MODE, true>: a fixed-length batch of reads with at least 32 values each leaves without any padding (a line shared by
// two reads is assembled at the end of the unit); batches with non-ACGT reads (the ASCII side launch rewrites runs in place) and tile
// batches keep the line-padded runs
// MEASURED AND REJECTED (round 4, NOTEBOOK): 11 % fewer bytes written, and 10 % slower -- the kernel is not bound by the bytes it writes
// (without the shared lines, i.e. 12 % fewer lines, it takes exactly the padded kernel's time), and assembling the shared lines costs
// what it costs.  Built only with make EXPERIMENTS=1 and chosen only with BSK_COMPACT=1 (tests/test_gpu_compact_streams.py).
static bool stream_compact_ok(const bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p) {
#ifdef BSK_EXPERIMENTS
    if (!ctx->opt.compact || !b->uniform_len || b->alias || b->n_nonacgt || p->circular) return false;
    return b->uniform_len >= (u32)p->k + 31u;
#else
    (void)ctx, (void)b, (void)p;
    return false;
#endif
}

int make_plan_enc(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p, Plan &pl, bool use_ascii) {
    pl.nunits = (u32)((b->n + 63) / 64);
    int per_cu = 1;
    if (p->kind == BSK_MINIMIZER) {
        // fast path: 2-bit input, a window size with a compiled specialisation, positions that fit 15 bits
        // windows that select more positions per read than the slab kernel stages (32): per-read slabs + mid-read flushes
        const double nwin = (double)b->maxlen - p->k - p->w + 2;
        const u64 dense_slab = (std::min<u64>((u64)std::max(nwin, 0.0), (u64)(std::max(nwin, 0.0) * PlannerTable::slab_sel_num / (p->w + 1.0)) + PlannerTable::slab_sel_pad) + 15) & ~(u64)15;
        // k_minimizer_seg: per-read slabs of the expected count + 30 % + 4 (150 bp, w = 11: 32 tuples), rounded to 64-byte pieces
        const double exp_sel = std::max(nwin, 0.0) * 2.0 / (p->w + 1.0) + 1.0;
        [[maybe_unused]] const u64 seg_slab = (std::min<u64>((u64)std::max(nwin, 1.0), (u64)(exp_sel * 1.3) + 4) + 7) & ~(u64)7;  // (make EXPERIMENTS=1)
        // unit rows through a ring (kernels_ring.hpp): reads that select more tuples than k_minimizer_pk stages (longer than ~156 bases at
        // w = 11) up to the length where the lanes of a unit drift too far apart for a ring of 16 rows (measured: DESIGN.md 3.2)
        const double exp_tuples = nwin * 2.0 / (p->w + 1.0);
        // (measured, profiles/r05/pkd_ring_sweep.txt: w = 3..13 x 100..450 bases against k_minimizer_pkd, which holds 720-820 Gbases/s at any length
        // (w >= 9) where the unit-row kernel falls with it as its lanes drift apart.  The crossover in expected tuples per read: 72 / 80 / 80 / 65
        // at w = 3 / 4 / 5 / 6 -- two or more blocks per flush round there, and the packed machine's per-block overhead weighs more on short
        // blocks --, 40 at w = 7, and 41 / 42 / 45 / 46 / 50 / 53 at w = 8 .. 13 in that sweep, 5 % of k_minimizer_pkd's rate lower since its
        // flush rounds are two blocks there: 17 + 2.5 w.  Round 4's rule, 34 + 2 w, was fitted against k_minimizer_dense.  BSK_RING_MAX overrides.)
        const double ring_cap = ctx->opt.ring_max ? (double)ctx->opt.ring_max : PlannerTable::ring_cap[std::min(std::max(p->w, 0), 13)];  // (planner_table.hpp: fitted per w by scripts/fit_planner.py)
        const double ring_sel = ctx->opt.ring_sel10 ? ctx->opt.ring_sel10 / 10.0 : (double)PlannerTable::slab_sel_num;  // (dev: BSK_RING_SEL10)
        const bool ring_wins = ctx->opt.ring ? true : exp_tuples > (double)ctx->opt.dense_min && exp_tuples <= ring_cap;
#ifdef BSK_EXPERIMENTS  // the two measured-and-rejected minimizer kernels (make EXPERIMENTS=1; NOTEBOOK round 2): never planned without their switch
        if (!use_ascii && p->w == 11 && p->k + p->w <= 65 && b->maxlen < 32768u && nwin >= 1.0 && ctx->opt.wpr && !ctx->no_dense && slab_budget_ok(b, seg_slab) &&
            !ctx->opt.force_generic) {
            pl.which = K_MIN_WPR;  // the A/B experiment: one read per wavefront (kernels_wpr.hpp); a ticket is 64 reads
            pl.fast_w = p->w;
            pl.slab = true;
            pl.slab_read = seg_slab;
            pl.slab_unit = 64 * pl.slab_read;
            pl.slab_total = (u64)pl.nunits * pl.slab_unit;
            per_cu = wpr_minimizer_blocks_per_cu();
        } else
        if (!use_ascii && seg_minimizer_supported(p->w) && b->maxlen < 32768u && nwin >= 1.0 && ctx->opt.seg && !ctx->no_dense && slab_budget_ok(b, seg_slab) &&
            !ctx->opt.force_generic) {
            pl.which = K_MIN_SEG;
            pl.fast_w = p->w;
            pl.slab = true;
            pl.slab_read = seg_slab;
            pl.slab_unit = 64 * pl.slab_read;
            pl.slab_total = (u64)pl.nunits * pl.slab_unit;
            per_cu = seg_minimizer_blocks_per_cu(p->w);
        } else
#endif
        if (!use_ascii && ring_minimizer_supported(p->w) && !b->alias && nwin >= 1.0 && nwin < (double)PlannerTable::ring_nwin_max && ring_wins && !ctx->opt.force_generic && !ctx->opt.no_ring &&
            !ctx->no_syn_pk && slab_budget_ok(b, ring_rows(nwin, p->w, ring_sel))) {
            pl.which = K_MIN_RING;  // w <= 13: packed window machine, unit rows through a ring of staged rows (kernels_ring.hpp)
            pl.fast_w = p->w;
            pl.fast_k = b->maxlen > ring_minimizer_short_bases() ? 1 : 0;  // (the kernel's LONG argument, for plan_name)
            pl.slab = true;
            pl.slab_unit = (u64)64 * ring_rows(nwin, p->w, ring_sel);
            pl.slab_total = (u64)pl.nunits * pl.slab_unit;
            pl.bin_gran = bin_gran_for(ctx, b, p->w);
            per_cu = ring_minimizer_blocks_per_cu(p->w);
        } else
        if (!use_ascii && pkd_minimizer_supported(p->w) && b->maxlen < 32768u && nwin * 2.0 / (p->w + 1.0) > (double)ctx->opt.dense_min && !ctx->no_dense &&
            !ctx->no_syn_pk && slab_budget_ok(b, dense_slab) && !ctx->opt.force_generic && !ctx->opt.no_dense && !ctx->opt.no_pk && !ctx->opt.no_pkd) {
            pl.which = K_MIN_PKD;  // w <= 13: the packed window machine over per-read slabs and mid-read flushes (kernels_pkd.hpp)
            pl.fast_w = p->w;
            pl.slab = true;
            pl.slab_read = dense_slab;  // whole 128-byte lines of hashes per read
            pl.slab_unit = 64 * pl.slab_read;
            pl.slab_total = (u64)pl.nunits * pl.slab_unit;
            pl.bin_gran = bin_gran_for(ctx, b, p->w);
            per_cu = pkd_minimizer_blocks_per_cu(p->w);
        } else
        if (!use_ascii && dense_minimizer_supported(p->w) && b->maxlen < 32768u && nwin * 2.0 / (p->w + 1.0) > (double)ctx->opt.dense_min && !ctx->no_dense && slab_budget_ok(b, dense_slab) &&
            !ctx->opt.force_generic && !ctx->opt.no_dense) {
            pl.which = K_MIN_DENSE;
            pl.fast_w = p->w;
            pl.slab = true;
            pl.slab_read = std::min<u64>((u64)nwin, (u64)(nwin * PlannerTable::slab_sel_num / (p->w + 1.0)) + PlannerTable::slab_sel_pad);
            pl.slab_read = (pl.slab_read + 15) & ~(u64)15;  // whole 128-byte lines of hashes per read
            pl.slab_unit = 64 * pl.slab_read;
            pl.slab_total = (u64)pl.nunits * pl.slab_unit;
            pl.bin_gran = bin_gran_for(ctx, b, p->w);
            per_cu = dense_minimizer_blocks_per_cu(p->w);
        } else if (!use_ascii && pk_minimizer_supported(p->w) && b->maxlen < 32768u && !ctx->opt.force_generic && !ctx->opt.no_pk && !ctx->no_syn_pk) {
            pl.which = K_MIN_PK;  // w <= 16: packed 32-bit window machine (kernels_pk.hpp)
            pl.fast_w = p->w;
            pl.fast_k = b->maxlen > pk_minimizer_short_bases() ? 1 : 0;  // (the kernel's LONG argument, for plan_name)
            pl.slab = true;
            pl.slab_unit = (u64)64 * BSK_FAST_CAP;
            pl.slab_total = (u64)pl.nunits * pl.slab_unit;
            pl.bin_gran = bin_gran_for(ctx, b, p->w);
            per_cu = pk_minimizer_blocks_per_cu(p->w);
        } else if (!use_ascii && fast_minimizer_supported(p->w) && b->maxlen < 32768u && !ctx->opt.force_generic) {
            pl.which = K_MIN_FAST;
            pl.fast_w = p->w;
            pl.slab = true;
            pl.slab_unit = (u64)64 * BSK_FAST_CAP;
            pl.slab_total = (u64)pl.nunits * pl.slab_unit;
            per_cu = fast_minimizer_blocks_per_cu(p->w);
        } else {
            pl.which = use_ascii ? K_MIN_GEN_A : K_MIN_GEN_P;
            per_cu = use_ascii ? occ(OCC_MIN_GEN_A) : occ(OCC_MIN_GEN_P);
            pl.ring_w = (u32)p->w;
        }
    } else if (p->kind == BSK_NTHASH) {
        if (!use_ascii && b->maxlen + (u32)(p->circular ? p->k : 0) <= 16u * (BSK_NT_FAST_WORDS - 2) && !ctx->opt.force_generic) {
            pl.which = K_NT_FAST;
            // write-bound kernel: measured fastest at 4 waves/CU (more concurrent 128-byte write streams per XCD
            // thrash the L2 write-combining: 3.65 ms vs 5.27 ms at 15 waves/CU for 10M reads)
            pl.compact = stream_compact_ok(ctx, b, p);
            per_cu = p->canonical ? occ(OCC_NT_FAST1) : occ(OCC_NT_FAST0);
#ifdef BSK_EXPERIMENTS
            if (pl.compact) per_cu = p->canonical ? occ(OCC_NT_FAST1C) : occ(OCC_NT_FAST0C);
#endif
        } else {
            pl.which = use_ascii ? K_NT_A : K_NT_P;
            per_cu = use_ascii ? occ(OCC_NT_A) : occ(OCC_NT_P);
        }
    } else if (p->kind == BSK_SYNCMER) {
        // packed machine: reads whose words fit a lane's registers, and few enough selections that a pair of reads stages in the
        // kernel's short columns (expected 1.5 / (k-s+1) of the windows: 7.1 of 101 at k=31 s=11, 150 bp, measured).  Two rows of slack
        // (round 4, scripts/dev/perf_syn_len.py: with six, reads of 165..188 bases ran on k_syncmer_fast at 640 instead of 800-850
        // Gbases/s; with none, 195-base reads fill their columns, list a quarter of the batch and fall back after a wasted run)
        const double syn_nwin = (double)b->maxlen - 2.0 * p->k + p->s + 2.0;
        const double syn_rows = 2.0 * (syn_nwin * PlannerTable::syn_sel_num / (p->k - p->s + 1.0) + 0.5) + (double)ctx->opt.syn_margin;
        auto syn_pk_fits = [&](bool lng) {
            // (the long plan: an eighth more than the short plan's rule, the spread of a pair's count grows with the count.  k=31 s=11,
            // scripts/dev/perf_syn_long.py: 250 / 300 / 350 / 380-base reads 818 / 750 / 759 / 680 Gbases/s -- at 380 the columns begin
            // to fill -- against 635 / 597 / 604 / 416 on k_syncmer_fast; 400-base reads want 59.3 of the 58 rows and stay there)
            const double want = lng ? syn_rows + PlannerTable::syn_long_spread * (syn_rows - (double)ctx->opt.syn_margin) : syn_rows;
            return pk_syncmer_supported(p->k - p->s, lng) && b->maxlen <= pk_syncmer_max_bases(lng) && want <= (double)pk_syncmer_pair_rows(lng);
        };
        const bool syn_short = syn_pk_fits(false), syn_lng = !syn_short && !ctx->opt.no_syn_long && syn_pk_fits(true);
        // small s: equal s-mers inside one 2w window are the rule (s = 7: 8 192 canonical values, half of the 150-base reads hold such a
        // pair), every such read is the exact machine's, the list (a quarter of the batch) fills up and the call falls back after a
        // wasted run.  Expected pairs per read = windows x 2w x 2 / 4^s; beyond 0.2 the packed kernels are not planned.
        const bool syn_ties = std::max(syn_nwin, 0.0) * 4.0 * (p->k - p->s) / std::pow(4.0, (double)std::min(p->s, 24)) > PlannerTable::syn_tie_pairs_max;
#ifdef BSK_EXPERIMENTS
        // the two-pass plan (kernels_syncmer_sel.hpp): selection by the packed s-mer machine, then ONLY the selected k-mers are hashed -- no
        // staging columns, so neither the rows-per-pair rule above nor column overflows apply: any read whose words fit the registers.
        // MEASURED AND NOT PLANNED (round 5, NOTEBOOK 5.4): the selection pass alone runs at 1 575 Gbases/s, but the second pass is bound by
        // the latency of its loads behind its stores (72 % of its wave cycles wait) and the two together reach 870 against k_syncmer_pk's
        // 950-966.  Built with make EXPERIMENTS=1, chosen with BSK_SYN_SEL=1 (tests/test_gpu_experiments.py keeps it exact).
        if (!use_ascii && ctx->opt.syn_sel && fast_syncmer_supported(p->k, p->s) && sel_syncmer_supported(p->k - p->s) && b->maxlen <= sel_syncmer_max_bases() && p->k <= 64 && !syn_ties &&
            !ctx->opt.force_generic && !ctx->opt.no_pk && !ctx->no_syn_pk && b->n < (1ULL << 32)) {
            pl.which = K_SYN_SEL;
            pl.fast_w = p->k - p->s;
            pl.slab = true;   // (the slab plumbing: [0, slab_total) is the DENSE region of the unlisted reads, the overflow region behind it the listed reads')
            pl.slab_unit = 0;
            // expected 1.5 / (k - s + 1) of the windows (7.1 of 101 at k = 31, s = 11: measured) + 25 %; an undershoot is seen by pass 2 and the call is sized again
            const double per_read = std::max(syn_nwin, 1.0) * 1.5 / (p->k - p->s + 1.0) * 1.25 + 2.0;
            pl.slab_total = ((u64)((double)b->n * std::min(per_read, std::max(syn_nwin, 1.0))) + 4096 + 63) & ~(u64)63;
            if (ctx->sel_need > pl.slab_total) pl.slab_total = (ctx->sel_need + 63) & ~(u64)63;  // (set while a call is being sized again)
            pl.bin_gran = bin_gran_for(ctx, b, p->k - p->s);
            per_cu = sel_syncmer_blocks_per_cu(pl.fast_w);
        } else
#endif
        // the fused-emit kernel (round 6, kernels_syncmer_pf.hpp): the s-mer machine alone + from-scratch hashes of what was selected at the
        // end of every unit -- no staging columns, so the rows-per-pair rule above does not apply: any read whose words fit a lane's
        // registers, whose blocks fit the mask rows, k <= 64 (the emit's window) and <= BSK_PF_TCAP / 64 expected selections per read
        const u32 syn_ns_max = b->maxlen >= (u32)p->s ? b->maxlen - (u32)p->s + 1u : 0u;
        auto syn_pf_fits = [&](bool lng) {  // (expected selections per read with a seventh of room below what a unit's emit phase takes)
            const double dens = std::max(syn_nwin, 0.0) * PlannerTable::syn_sel_num / (p->k - p->s + 1.0);
            const double dmax = ctx->opt.pf_density ? (double)ctx->opt.pf_density : (double)pf_syncmer_unit_tuples(lng) / 64.0 * PlannerTable::pf_list_fill;
            return !ctx->opt.no_syn_pf && pf_syncmer_supported(p->k - p->s, lng) && b->maxlen <= pf_syncmer_max_bases(lng) && p->k <= PlannerTable::pf_k_max &&
                   (syn_ns_max + (u32)(p->k - p->s) - 1u) / (u32)(p->k - p->s) <= pf_syncmer_mask_rows(lng) + 1u && dens <= dmax;
        };
        const bool syn_pf_short = syn_pf_fits(false), syn_pf_long = !syn_pf_short && !ctx->opt.no_syn_long && syn_pf_fits(true);
        const bool syn_pf = syn_pf_short || syn_pf_long;
        if (!use_ascii && fast_syncmer_supported(p->k, p->s) && (syn_short || syn_lng || syn_pf) && !syn_ties && !ctx->opt.force_generic && !ctx->opt.no_pk && !ctx->no_syn_pk) {
            pl.syn_fused = syn_pf;
            pl.syn_long = syn_pf ? syn_pf_long : syn_lng;
            pl.which = K_SYN_PK;
            pl.fast_w = p->k - p->s;
            pl.slab = true;
            pl.slab_unit = (u64)64 * BSK_SYN_CAP;
            pl.slab_total = (u64)pl.nunits * pl.slab_unit;
            pl.bin_gran = bin_gran_for(ctx, b, p->k - p->s);
            per_cu = pl.syn_fused ? pf_syncmer_blocks_per_cu(pl.fast_w, pl.syn_long) : pk_syncmer_blocks_per_cu(pl.fast_w, pl.syn_long);
        } else if (!use_ascii && fast_syncmer_supported(p->k, p->s) && b->maxlen < 32768u && !ctx->opt.force_generic) {
            pl.which = K_SYN_FAST;
            pl.fast_w = p->k - p->s;
            pl.slab = true;
            pl.slab_unit = (u64)64 * BSK_SYN_CAP;
            pl.slab_total = (u64)pl.nunits * pl.slab_unit;
            pl.bin_gran = bin_gran_for(ctx, b, p->k - p->s);
            per_cu = fast_syncmer_blocks_per_cu(pl.fast_w);
        } else {
            pl.which = use_ascii ? K_SYN_A : K_SYN_P;
            per_cu = use_ascii ? occ(OCC_SYN_A) : occ(OCC_SYN_P);
            pl.ring_w = (u32)std::max(1, 2 * (p->k - p->s));
        }
    } else if (p->kind == BSK_KMER) {
        if (!use_ascii && p->canonical > 0 && b->maxlen + (u32)(p->circular ? p->k : 0) <= 16u * (BSK_NT_FAST_WORDS - 2) &&
            !ctx->opt.force_generic) {
            pl.which = K_NT_FAST;  // same streaming kernel, MODE 2
            pl.compact = stream_compact_ok(ctx, b, p);
            per_cu = occ(OCC_NT_FAST2);
#ifdef BSK_EXPERIMENTS
            if (pl.compact) per_cu = occ(OCC_NT_FAST2C);
#endif
        } else if (!use_ascii && p->canonical <= 0 && !p->circular && b->maxlen <= 16u * (BSK_NT_FAST_WORDS - 2) && !ctx->opt.force_generic) {
            pl.which = K_NT_FAST;  // MODE 3: both strands (iterator.go:713-723), MODE 4: forward codes only (the tile pass of two-strand long sequences)
            per_cu = occ(p->canonical < 0 ? OCC_NT_FAST4 : OCC_NT_FAST3);
        } else {
            pl.which = use_ascii ? K_KMER_A : K_KMER_P;
            per_cu = use_ascii ? occ(OCC_KMER_A) : occ(OCC_KMER_P);
        }
    } else if (p->kind == BSK_SIMHASH) {
        const int nh = p->k - p->m + 1;
        if (!use_ascii && nh <= 63 && b->maxlen + (u32)(p->circular ? p->k : 0) <= 16u * (BSK_NT_FAST_WORDS - 2) &&
            !ctx->opt.force_generic) {
            pl.which = K_SIM_FAST;  // bit-sliced counters: 5 planes count to 31, 6 to 63
            pl.fast_w = nh <= 31 ? 5 : 6;
            const u32 ext_len = b->maxlen + (u32)(p->circular ? p->k : 0);
            pl.fast_k = ext_len <= 16u * (BSK_SIM_SHORT_WORDS - 2) ? 1 : ext_len <= 16u * (BSK_SIM_MID_WORDS - 2) ? 2 : 0;  // shorter reads: less LDS, more waves
            if (pl.fast_k == 1) per_cu = nh <= 31 ? occ(OCC_SIMF_5S) : occ(OCC_SIMF_6S);
            else if (pl.fast_k == 2) per_cu = nh <= 31 ? occ(OCC_SIMF_5M) : occ(OCC_SIMF_6M);
            else per_cu = nh <= 31 ? occ(OCC_SIMF_5) : occ(OCC_SIMF_6);
        } else {
            pl.which = use_ascii ? K_SIM_A : K_SIM_P;
            per_cu = use_ascii ? occ(OCC_SIM_A) : occ(OCC_SIM_P);
            pl.ring_w = (u32)nh;
        }
    } else if (p->kind == BSK_PROT_HASH) {
        if (b->alphabet == BSK_ALPHA_DNA) {  // the fused plan (sketch_impl checked that it applies)
            pl.which = K_PROT_HASH_FAST;
            pl.fused_dna = true;
            pl.fast_k = p->k;
            per_cu = fast_prot_hash_dna_blocks_per_cu(p->k);
        } else if (fast_prot_hash_supported(p->k) && !ctx->opt.force_generic) {
            pl.which = K_PROT_HASH_FAST;
            pl.fast_k = p->k;
            per_cu = fast_prot_hash_blocks_per_cu(p->k);
        } else {
            pl.which = K_PROT_HASH;
            per_cu = occ(OCC_PROT_HASH);
        }
    } else if (p->kind == BSK_PROT_MINIMIZER) {
        // a DNA batch here means the fused plan (sketch_impl checked that it applies): lengths in residues
        const u32 plen = b->alphabet == BSK_ALPHA_DNA ? (u32)translated_len(b->maxlen, 1) : b->maxlen;
        if (b->alphabet == BSK_ALPHA_DNA && ctx->no_prot_fast) return BSK_REPLAN_UNFUSED;  // slabs too small / too large: sketch_impl translates first
        if (b->alphabet == BSK_ALPHA_DNA || (fast_prot_supported(p->w, p->k) && b->maxlen < 65536u && b->maxlen >= (u32)(p->k + p->w) && !ctx->opt.force_generic &&
            !ctx->no_prot_fast && slab_budget_ok(b, (u64)b->maxlen))) {
            pl.which = K_PROT_MIN_FAST;
            pl.fused_dna = b->alphabet == BSK_ALPHA_DNA;
            pl.fast_w = p->w;
            pl.fast_k = p->k;
            pl.slab = true;
            const u64 nwin = plen >= (u32)(p->k + p->w) ? (u64)plen - p->k - p->w + 2 : 1;
            // mean 2/(w+1) of the windows, +30 % + 16 (+ 8 until round 5: 2 10^7 sequences of 100 residues at k = 8 w = 8 -- 19 +- 3 tuples in a
            // slab of 32 -- had one sequence over, and the whole batch fell back to the general kernel: 72 instead of 530 G residues/s)
            pl.slab_read = std::min<u64>(nwin, (u64)(nwin * PlannerTable::slab_sel_num / (p->w + 1.0)) + PlannerTable::slab_sel_pad);
            pl.slab_read = (pl.slab_read + 15) & ~(u64)15;  // whole 128-byte lines of hashes per sequence
            pl.slab_unit = 64 * pl.slab_read;
            pl.slab_total = (u64)pl.nunits * pl.slab_unit;
            per_cu = fast_prot_blocks_per_cu(p->w, p->k);
        } else {
            pl.which = K_PROT_MIN;
            per_cu = occ(OCC_PROT_MIN);
            pl.ring_w = (u32)p->w;
        }
    } else {
        ctx->err = "unknown kind";
        return BSK_ERR_ARG;
    }
    if (ctx->opt.waves_per_cu) per_cu = (int)ctx->opt.waves_per_cu;  // dev: occupancy experiments
    pl.grid = (int)std::max<u64>(1, std::min<u64>((u64)ctx->cus * per_cu, pl.nunits));
    pl.ring_entries = (size_t)pl.grid * pl.ring_w * 64;
    return BSK_OK;
}


// name of the planned kernel, as rocprofv3 shows it (bsk_result_plan; bench.py's roofline.kernel)
void plan_name(const Plan &pl, const bsk_params *p, bool tiled, int cus, bsk_result *res) {
    char b[80];
    switch (pl.which) {
        case K_MIN_GEN_P: snprintf(b, sizeof b, "k_minimizer_generic<0>"); break;
        case K_MIN_GEN_A: snprintf(b, sizeof b, "k_minimizer_generic<1>"); break;
        case K_NT_P: snprintf(b, sizeof b, "k_nthash_stream<0>"); break;
        case K_NT_A: snprintf(b, sizeof b, "k_nthash_stream<1>"); break;
        case K_MIN_FAST: snprintf(b, sizeof b, "k_minimizer_fast<%d,%d,true>", pl.fast_w, BSK_FAST_CAP); break;
        case K_MIN_PK: snprintf(b, sizeof b, "k_minimizer_pk<%d,%s>", pl.fast_w, pl.fast_k ? "true" : "false"); break;
        case K_MIN_RING: snprintf(b, sizeof b, "k_minimizer_ring<%d,%s>", pl.fast_w, pl.fast_k ? "true" : "false"); break;
        case K_MIN_DENSE: snprintf(b, sizeof b, "k_minimizer_dense<%d>", pl.fast_w); break;
        case K_MIN_PKD: snprintf(b, sizeof b, "k_minimizer_pkd<%d>", pl.fast_w); break;
        case K_MIN_SEG: snprintf(b, sizeof b, "k_minimizer_seg<%d>", pl.fast_w); break;
        case K_MIN_WPR: snprintf(b, sizeof b, "k_minimizer_wpr<%d>", pl.fast_w); break;
        case K_NT_FAST: snprintf(b, sizeof b, pl.compact ? "k_nthash_fast<%d,true>" : "k_nthash_fast<%d>", p->kind == BSK_KMER ? (p->canonical > 0 ? 2 : p->canonical < 0 ? 4 : 3) : p->canonical ? 1 : 0); break;
        case K_SYN_P: snprintf(b, sizeof b, "k_syncmer<0>"); break;
        case K_SYN_A: snprintf(b, sizeof b, "k_syncmer<1>"); break;
        case K_KMER_P: snprintf(b, sizeof b, "k_kmer<0>"); break;
        case K_KMER_A: snprintf(b, sizeof b, "k_kmer<1>"); break;
        case K_SIM_P: snprintf(b, sizeof b, "k_simhash<0>"); break;
        case K_SIM_A: snprintf(b, sizeof b, "k_simhash<1>"); break;
        case K_PROT_HASH: snprintf(b, sizeof b, "k_prot_hash"); break;
        case K_PROT_MIN: snprintf(b, sizeof b, "k_prot_minimizer"); break;
        case K_SYN_FAST: snprintf(b, sizeof b, "k_syncmer_fast<%d>", pl.fast_w); break;
        case K_SYN_PK: snprintf(b, sizeof b, pl.syn_fused ? (pl.syn_long ? "k_syncmer_pfl<%d>" : "k_syncmer_pf<%d>") : pl.syn_long ? "k_syncmer_pkl<%d>" : "k_syncmer_pk<%d>", pl.fast_w); break;
        case K_SYN_SEL: snprintf(b, sizeof b, "k_syncmer_sel<%d> + k_syncmer_emit", pl.fast_w); break;
        case K_PROT_MIN_FAST: snprintf(b, sizeof b, "k_prot_minimizer_fast<%d,%d,%s>", pl.fast_w, pl.fast_k, pl.fused_dna ? "true" : "false"); break;
        case K_PROT_HASH_FAST: snprintf(b, sizeof b, "k_prot_hash_fast<%d,%s>", pl.fast_k, pl.fused_dna ? "true" : "false"); break;
        case K_SIM_FAST:
            snprintf(b, sizeof b, "k_simhash_fast<%d,%d>", pl.fast_w, pl.fast_k == 1 ? BSK_SIM_SHORT_WORDS : pl.fast_k == 2 ? BSK_SIM_MID_WORDS : BSK_NT_FAST_WORDS);
            break;
        default: snprintf(b, sizeof b, "?"); break;
    }
    snprintf(res->plan, sizeof res->plan, "%s%s%s%s", b, pl.bin_gran ? " (length-binned units)" : "", pl.mixed ? " + ASCII side launch" : "", tiled ? " (over tiles)" : "");
    res->plan_grid = pl.grid;
    res->plan_per_cu = cus > 0 ? (pl.grid + cus - 1) / cus : 0;
}

// k_syncmer_pk's / k_minimizer_pk's list of reads for the exact machine: room for a quarter of the batch (a batch with more falls back
// to k_syncmer_fast / k_minimizer_fast)
// (list_append, kernels_generic.hpp: one segment per workgroup of the launch -- at least 1 024 entries each, so that a small batch of
// nothing but low-complexity reads still fits its segments)
u64 syn_pk_fixcap(u64 n, int grid) { return std::max<u64>((u64)grid * 1024, (n / 4 + (u64)grid) / (u64)grid * (u64)grid); }

