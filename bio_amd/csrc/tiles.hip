// tiles.hip -- long sequences: cut into overlapping tiles, the ordinary kernels over the tiles, stitch (kernels_tile.hpp).
#include "host_internal.hpp"
#include "kernels_host.hpp"
#include "kernels_tile.hpp"

// ------------------------------------------------------------------------------------
// long sequences: tile, run the ordinary kernels over the tiles, stitch (kernels_tile.hpp)
// ------------------------------------------------------------------------------------
bool kind_tiles(const bsk_params *p) {
    switch (p->kind) {
        case BSK_NTHASH:
        case BSK_SIMHASH:
        case BSK_MINIMIZER: return true;
        case BSK_KMER: return true;  // two-strand mode (iterator.go:713-723): forward codes over tiles, then k_two_strand
        case BSK_SYNCMER: return true;            // (s == k, "every k-mer", runs as the w = 1 minimizer: sketch_tiled)
        case BSK_PROT_HASH:
        case BSK_PROT_MINIMIZER: return true;
        default: return false;
    }
}

// the planner would put fixed-length 2-bit reads of a fitting length on k_syncmer_pkl (make_plan, BSK_SYNCMER)
static bool syn_long_plan_ok(const bsk_ctx *ctx, const bsk_params *p) {
    return pk_syncmer_supported(p->k - p->s, true) && fast_syncmer_supported(p->k, p->s) && !ctx->opt.no_syn_long && !ctx->opt.no_pk && !ctx->opt.force_generic &&
           p->s >= 9;  // (small s: equal s-mers inside a window are the rule and the packed kernels are not planned)
}

// positions one tile owns: about 22 tuples per tile (the kernels stage 32 per lane; 16 for syncmers), a multiple of 16
static u32 tile_positions(const bsk_ctx *ctx, const bsk_params *p, u64 n_bases, u64 maxlen) {
    u32 tp;
    if (p->kind == BSK_MINIMIZER || p->kind == BSK_PROT_MINIMIZER) {
        tp = 16u * std::max<u32>(2, (u32)((double)PlannerTable::tile_min_tuples * (p->w + 1.0) / 2.0 / 16.0));
        // windows only k_minimizer_fast takes (w >= 17): its lanes stage in PAIRS of reads sharing a 56-row column, and a tile of 22 owned
        // tuples carries 25 with its overlap -- 50 +- 5 per pair, a tenth of the pairs over, i.e. every unit run again with direct stores
        // (w = 20, 700-base reads over such tiles: 288 Gbases/s).  20 expected tuples per tile instead.
        if (p->kind == BSK_MINIMIZER && p->w == 1) tp = 512;  // every position selected: k_minimizer_dense<1>'s per-read slabs take any tile, and 32 positions + k + 18 of overlap were two thirds overlap
        if (p->kind == BSK_MINIMIZER && !pk_minimizer_supported(p->w) && !dense_minimizer_supported(p->w)) {
            const double room = 10.0 * (p->w + 1.0) - p->w - 18.0;
            tp = 16u * std::max<u32>(2, (u32)(room / 16.0));
        }
        // round 5: a tile carries 2w + k + 16 bases of overlap, so 128 owned positions at k=21 w=11 are a 187-base tile that selects 26
        // tuples -- k_minimizer_dense's (526 Gbases/s of tile bases); tiles whose windows (tp + w + 18) stay at the packed machine's
        // tuple count run on k_minimizer_pk at twice that, which more than pays for the shorter tile (2 10^9 bases of long sequences:
        // 10.0 -> 8.8 ms, scripts/dev/perf_long2.py)
        if (p->kind == BSK_MINIMIZER && pk_minimizer_supported(p->w) && !ctx->opt.no_pk && !ctx->opt.force_generic) {
            const double room = (double)ctx->opt.dense_min * (p->w + 1.0) / 2.0 - p->w - 18.0;
            const u32 tpk = room > 0 ? 16u * (u32)(room / 16.0) : 0u;
            if (tpk >= 64u) tp = tpk;
            // k_minimizer_pkd (round 5) runs tiles of any length at 0.65 of k_minimizer_pk's rate, and a tile of 1 024 positions carries
            // 5 % of overlap instead of 38 %, a tenth of the tiles to cut, stitch and gather: 2 10^9 bases of long sequences 9.1 -> 7.4 ms,
            // 2 10^8 1.5 -> 1.2 ms with tiles of 512 -- as long as there are tiles enough for every lane of the device (scripts/dev/
            // run_tilepos.sh: with fewer than ~300 000 the larger tile loses: 2 10^7 bases 0.45 ms on 96-position tiles, 0.7 on 512)
            if (pkd_minimizer_supported(p->w) && !ctx->opt.no_pkd && !ctx->opt.no_dense && !ctx->no_dense && !ctx->no_syn_pk)
                for (u32 big = 1024; big >= 256; big >>= 1)
                    if (n_bases / big >= (u64)PlannerTable::tile_big_tiles_min) {
                        tp = big;
                        break;
                    }
        }
    }
    else if (p->kind == BSK_SYNCMER) {
        tp = 16u * std::max<u32>(2, (u32)((double)PlannerTable::tile_syn_tuples * (p->k - p->s + 1.0) / 2.0 / 16.0));
        // round 4: k_syncmer_pkl takes tiles three times as long at 0.9 of the rate, and a tile carries 3k + 16 bases of overlap: at k=31
        // s=11 tiles of 112 + 109 bases spend half of the kernel on overlaps, tiles of 224 + 109 a third (~21 expected selections
        // per tile: where the long plan's rate is still flat, scripts/dev/perf_syn_long.py)
        const int w = p->k - p->s;
        const long long lt = std::min<long long>(480, 14LL * (w + 1) + 2LL * p->k - p->s - 2), over = 3LL * p->k - 2LL * p->s + 12;  // (a tile's bases beyond its own positions: w - 1 idx before them, 2k - s - 1 after the last, up to 15 of alignment -- k_tile_desc)
        if (syn_long_plan_ok(ctx, p) && lt - over > (long long)tp) {
            tp = (u32)(lt - over) & ~15u;
            // (the tightest tiles are the longest the plan's columns take.  A wavefront runs 64 tiles in lock step, so the tiles of the batch's
            // longest read are made equal: 700 bases at k=31 s=11 are 224 + 224 + 203 positions, not 256 + 256 + 139; 420 bases 192 + 179 --
            // 224 + 147 ran 9 % faster than 256 + 115 there.  500 bases were 224 + 224 + 3 with the overlap priced at 3k + 16: 420 -> 650
            // Gbases/s; 1 000 / 3 000 bases 504 / 534 -> 566 / 560: scripts/dev/run_synlen.sh)
            const long long np = (long long)maxlen + p->s + 2 - 2LL * p->k;
            if (np > (long long)tp) {
                const long long nt = (np + tp - 1) / tp;
                const u32 bal = (u32)(((np + nt - 1) / nt + 15) & ~15LL);
                if (bal >= 32u && bal < tp) tp = bal;
            }
        }
    }
    else tp = 256;
    tp = std::min<u32>(tp, 8192);
    const u32 forced = ctx->opt.tile_pos;  // tests: exercise the tile seams
    if (forced) tp = std::max<u32>(16, (forced + 15) & ~15u);
    return tp;
}

int sketch_tiled(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p_in, int circ_ext, bsk_result **result, int warmup, int iters,
                        float *kernel_ms) {
    // a syncmer sketch with s == k yields every k-mer with its index (sketch.go:328-331) -- exactly the minimizer sketch with
    // w = 1 (sketch.go:218-222), and both refuse a sequence shorter than k (the syncmer through its hasher, sketch.go:179-182):
    // tiles run it as that
    bsk_params pw1 = *p_in;
    const bool syn_all = p_in->kind == BSK_SYNCMER && p_in->s == p_in->k;
    if (syn_all) {
        pw1.kind = BSK_MINIMIZER;
        pw1.w = 1;
    }
    const bool two_strand = p_in->kind == BSK_KMER && !p_in->canonical;
    if (two_strand) pw1.canonical = -1;  // internal: forward codes only (KArgs::one_strand)
    const bsk_params *p = &pw1;
    const u64 n = b->n;
    TileGeo geo;
    geo.kind = p->kind;
    geo.k = p->k;
    geo.w = p->kind == BSK_SYNCMER ? p->k - p->s : p->w;
    geo.s = p->s;
    geo.tp = tile_positions(ctx, p, b->n_bases, b->maxlen);
    // Dense tiles (round 6, kernels_minimizer_pf.hpp): the tile kernel writes the final tuples -- owned positions only, shifted, every unit
    // packed behind the one before through a decoupled look-back -- and no stitch pass runs.  Its tiles are sized by its own limits: 160
    // bases of a tile in LDS (tp + 2w + k + 16), 16 blocks of W k-mers (tp + 2w + 16 <= 16 w), 86 % of the emit list (64 tp 2 / (w + 1) <= 1 100).
    u32 dense_tp = 0;
    if (p->kind == BSK_MINIMIZER && !syn_all && pft_minimizer_supported(p->w) && p->k <= PlannerTable::pf_k_max && ctx->opt.tile_dense && !ctx->opt.force_generic &&
        !ctx->opt.no_pk && !ctx->opt.tile_pos) {
        const long long la = (long long)pft_minimizer_max_tile_bases() - 16 - 2LL * p->w - p->k, lb = ((long long)pft_minimizer_mask_rows() - 3) * p->w - 13,  // (nk <= tp + 2w + 14 k-mers in at most 16 blocks of w)
                        lc = (long long)((double)pft_minimizer_unit_tuples() * PlannerTable::pf_list_fill / 64.0 * (p->w + 1.0) / 2.0);
        const long long t = std::min(la, std::min(lb, lc)) & ~15LL;
        if (t >= 32) dense_tp = (u32)t;
    }
    geo.circ_ext = circ_ext;
    geo.syn_all = syn_all ? 1 : 0;
    const bool stream = !kind_has_pos(p->kind);
    const bool prot = b->alphabet == BSK_ALPHA_PROTEIN;
    SeqTab seq{b->desc, b->fw, b->llen, b->aoff, n, prot ? b->rflags : nullptr};  // protein rflags: the translate kernel's short flags
    u64 *tstart = nullptr, *oexcl = nullptr, *sbad = nullptr;
    u32 *sflags = nullptr;
    TileTab tt{nullptr, nullptr, nullptr, nullptr, nullptr};
    bsk_batch *tb = nullptr;
    bsk_result *tres = nullptr, *fin = nullptr;
    // temporaries come from the context's grow-only pool (slot numbers below); the tile-level result is cached there too
    auto pool = [&](int slot, size_t bytes, void **out) -> hipError_t {
        if (ctx->tmp_cap[slot] < bytes) {
            (void)hipFree(ctx->tmp[slot]);
            ctx->tmp[slot] = nullptr;
            ctx->tmp_cap[slot] = 0;
            const size_t want = bytes + bytes / 4 + 256;
            const hipError_t e = hipMalloc(&ctx->tmp[slot], want);
            if (e != hipSuccess) return e;
            ctx->tmp_cap[slot] = want;
        }
        *out = ctx->tmp[slot];
        return hipSuccess;
    };
    bsk_result *old = nullptr;  // the caller's previous result (below)
    auto done = [&](int code) {
        if (tb) {  // the tile batch only borrowed its descriptor / flag arrays
            tb->desc = nullptr;
            tb->adesc = nullptr;
            tb->rflags = nullptr;
            bsk_batch_destroy(tb);
        }
        if (code != BSK_OK && fin) bsk_result_release(fin);
        if (old) {
            bsk_result_release(old);
            old = nullptr;
        }
        if (ctx->opt.no_tile_cache && ctx->tile_res) {  // dev switch
            bsk_result_release(ctx->tile_res);
            ctx->tile_res = nullptr;
        }
        return code;
    };
    bsk_result *&tres_slot = ctx->tile_res;
#define TCHK(call)                                                  \
    do {                                                            \
        hipError_t e__ = (call);                                    \
        if (e__ != hipSuccess) return done(fail_hip(ctx, e__, #call)); \
    } while (0)
    const bool timing = ctx->opt.timing;  // dev: wall time of the phases of a tiled call, to stderr
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(ctx->stream);
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[tiled] %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    // the caller's previous result: its arrays serve again where they fit (a timed re-run, a class plan's part on every launch, a streaming
    // caller's next chunk: hipFree synchronises the whole device and five hipMalloc per call cost more than a small part's kernels)
    old = *result;
    *result = nullptr;
    auto drop_old = [&]() {
        if (old) bsk_result_release(old);
        old = nullptr;
    };
    if (old && (old->ctx != ctx || !old->wfirst || old->n != n || old->arrays_borrowed || old->classes || !kind_has_pos(p_in->kind) || two_strand || old->kind != p_in->kind)) drop_old();
    // 1. tiles per sequence -> first tile of every sequence
    const u32 nunits = (u32)((n + 63) / 64);
    int rc = ensure_scratch(ctx, std::max<u32>(nunits, 1), 0);
    if (rc != BSK_OK) return done(rc);
    TCHK(pool(0, (n + 1) * 8, (void **)&tstart));
    TCHK(hipMemsetAsync(tstart, 0, (n + 1) * 8, ctx->stream));
    // Without host round trips (round 6): the number of tiles is bounded on the host -- a sequence of L bases has at most L positions, so
    // at most L / tp + 1 tiles -- every array and grid is sized by the bound, the entries beyond the true count (on the device: tstart[n])
    // are empty tiles, the tile kernels launch ONCE into slabs sized by the plan (run_planned, ctx->defer) and the only synchronisation is
    // the call's last one, which also brings the overflow flags: a call that finds one set runs again the old way (tile_sync).  Batches
    // with a non-ACGT letter (per-tile flags pick the side launch's tiles), proteins and the two-strand k-mer mode keep the round trips.
    const bool defer = !ctx->opt.no_tile_defer && !ctx->tile_sync && !prot && b->n_nonacgt == 0 && !two_strand && n > 0;
    const bool dense = defer && dense_tp && !ctx->tile_async && !circ_ext;
    if (dense) geo.tp = dense_tp;
    const bool async_final = defer && ctx->tile_async && warmup + iters == 0;
    ctx->tile_was_async = async_final;
    u64 nt = 0;
    if (n) {
        TCHK(hipMemsetAsync(ctx->d_ticket, 0, 2 * sizeof(u32), ctx->stream));
        TCHK(hipMemsetAsync(ctx->d_lookback, 0, (size_t)nunits * 8, ctx->stream));
        TileArgs ta{seq, geo, nunits, tstart, ctx->d_ticket, ctx->d_lookback};
        hipLaunchKernelGGL(k_tile_count, dim3(std::min<u32>(nunits, (u32)ctx->cus * 8)), dim3(64), 0, ctx->stream, ta);
        TCHK(hipGetLastError());
        if (defer) {
            nt = b->n_bases / geo.tp + n;
        } else {
            TCHK(hipMemcpyAsync(ctx->h_pinned, tstart + n, 8, hipMemcpyDeviceToHost, ctx->stream));
            TCHK(hipStreamSynchronize(ctx->stream));
            nt = ctx->h_pinned[0];
        }
    }
    lap("tile count");
    // 2. tile table + a batch whose "reads" are the tiles (aliases the words / bytes of b)
    const bool use_ascii = prot || b->n_nonacgt > 0;  // residues are bytes
    const size_t nta = nt ? nt : 1;
    TCHK(pool(1, nta * 8, (void **)&tt.desc));
    if (use_ascii) TCHK(pool(2, nta * 8, (void **)&tt.adesc));
    TCHK(pool(3, nta * 4, (void **)&tt.seq));
    TCHK(pool(4, nta * 8, (void **)&tt.shift));
    TCHK(pool(5, nta * 8, (void **)&tt.keep));
    u8 *tflags = nullptr;  // per tile: holds a non-ACGT letter (from the per-word bits of the batch); owned by tb later
    if (prot || (use_ascii && b->wbits)) {  // protein: all-zero flags = "the input-length rule was already applied" for every tile
        TCHK(pool(6, nta, (void **)&tflags));
        TCHK(hipMemsetAsync(tflags, 0, nta, ctx->stream));
    }
    if (nt) {
        hipLaunchKernelGGL(k_tile_build, dim3(grid_for(ctx, nt, 256)), dim3(256), 0, ctx->stream, seq, geo, tstart, nt, tt, b->wbits,
                           prot ? nullptr : tflags);
        TCHK(hipGetLastError());
    }
    u64 n_bad_tiles = tflags ? 0 : b->n_nonacgt;  // with per-tile flags: counted below (no tiles, no flagged tiles)
    if (tflags && nt && !prot) {
        TCHK(hipMemsetAsync(ctx->d_ticket, 0, 2 * sizeof(u32), ctx->stream));
        hipLaunchKernelGGL(k_count_flags, dim3(grid_for(ctx, nt, 256)), dim3(256), 0, ctx->stream, tflags, nt, ctx->d_ticket + 1);
        TCHK(hipMemcpyAsync(ctx->h_pinned, ctx->d_ticket, 2 * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
        TCHK(hipStreamSynchronize(ctx->stream));
        n_bad_tiles = ((u32 *)ctx->h_pinned)[1];
    }
    tb = new (std::nothrow) bsk_batch();
    if (!tb) return done(BSK_ERR_NOMEM);
    tb->ctx = ctx;
    tb->alphabet = b->alphabet;
    tb->pairs = b->pairs;
    tb->alias = true;
    tb->n = nt;
    tb->words = b->words;
    tb->ascii = b->ascii;
    tb->desc = tt.desc;
    tb->adesc = tt.adesc;
    tb->n_nonacgt = n_bad_tiles;
    tb->rflags = tflags;  // NULL: no per-tile knowledge, every tile runs on the ASCII kernels
    if (prot) tb->n_nonacgt = 0;
    if (!prot && tflags && n_bad_tiles && nt < (1ULL << 32)) {
        rc = build_subset(ctx, tb);
        if (rc != BSK_OK) return done(rc);
    }
    const u64 over = (p->kind == BSK_MINIMIZER || p->kind == BSK_PROT_MINIMIZER) ? 2ULL * p->w + p->k + 16
                     : p->kind == BSK_SYNCMER                                     ? 3ULL * p->k - 2ULL * p->s + 12  // (k_tile_desc: w - 1 idx before the tile's positions, 2k - s - 1 bases after the last, <= 15 of alignment; 3k + 16 kept tiles of k - s < 20 off the long packed plan once tile_positions sized them by the exact extent)
                                                                                  : (u64)p->k;
    tb->maxlen = (u32)std::min<u64>((u64)geo.tp + over, (u64)b->maxlen);
    tb->n_bases = nt * tb->maxlen;  // upper bound: sizes the first capacity guess
    tb->n_words = b->n_words;
    bsk_params p2 = *p;
    p2.circular = 0;
    lap("tile table");
    // 3. the ordinary kernels over the tiles
    // the cached tile result belongs to an earlier batch: always size (one untimed run) before any timed repetition
    if (dense) {
        // 3d. dense tiles: the tile kernel writes the FINAL tuples into the sequence result's own arrays (expected 2 / (w + 1) per position + a
        // quarter; a batch that selects more -- long low-complexity stretches -- raises the flag and the call runs again the old way)
        const u64 need = (u64)((double)b->n_bases * 2.0 / (p->w + 1.0) * 1.25) + (1u << 20);
        if (old && old->hash && old->pos && old->alloc_cap >= need) {
            fin = old;
            old = nullptr;
        } else {
            drop_old();
            fin = new (std::nothrow) bsk_result();
            if (!fin) return done(BSK_ERR_NOMEM);
            fin->ctx = ctx;
            fin->n = n;
            fin->kind = p_in->kind;
            fin->has_pos = 1;
            TCHK(hipMalloc(&fin->status, n ? n : 1));
            TCHK(hipMalloc(&fin->wfirst, (n ? n : 1) * 8));
            TCHK(hipMalloc(&fin->wcount, (n ? n : 1) * 8));
            TCHK(spare_take(ctx, (void **)&fin->hash, need * 8));  // (a released result's arrays, if the context holds ones that fit: bsk_ctx::spare)
            TCHK(spare_take(ctx, (void **)&fin->pos, need * 4));
            fin->cap = fin->alloc_cap = need;
        }
        rc = result_prepare(ctx, &tres_slot, nt, p->kind, 0);  // (reference words and status bytes per tile; the tuples are the sequence result's)
        if (rc != BSK_OK) return done(rc);
        tres = tres_slot;
        const u32 tunits = (u32)((nt + 63) / 64);
        // scratch of the units' prefix (kernels_minimizer_pf.hpp): a look-back granule per chunk of 64 units, a total per unit, then the eight
        // ticket heads (one per XCD, 128 B apart)
        const size_t lb_entries = pft_minimizer_scratch_words(tunits);
        rc = ensure_scratch(ctx, lb_entries, 0);
        if (rc != BSK_OK) return done(rc);
        const int per_cu = pft_minimizer_blocks_per_cu(p->w);
        const int grid = (int)std::max<u64>(1, std::min<u64>((u64)ctx->cus * per_cu, tunits));
        KArgs ka;
        memset(&ka, 0, sizeof ka);
        ka.words = b->words;
        ka.desc = tt.desc;
        ka.n = nt;
        ka.nunits = tunits;
        ka.kind = p->kind;
        ka.k = p->k;
        ka.w = p->w;
        ka.refs = tres->refs;
        ka.status = tres->status;
        ka.hash = fin->hash;
        ka.pos = fin->pos;
        ka.cap = fin->alloc_cap;
        ka.ticket = ctx->d_ticket;
        ka.lookback = ctx->d_lookback;
        ka.tkeep = tt.keep;
        ka.tshift = tt.shift;
        ka.len_mask = 0xffffffu;
        std::vector<hipEvent_t> evs;
        for (int i = 0; kernel_ms && i < 2 * iters; ++i) {
            hipEvent_t e = nullptr;
            (void)hipEventCreate(&e);
            evs.push_back(e);
        }
        for (int it = -1; it < warmup + iters; ++it) {  // (-1: the call's own run)
            if (it >= 0 && warmup + iters == 0) break;
            TCHK(hipMemsetAsync(ctx->d_ticket, 0, 8 * sizeof(u32), ctx->stream));
            TCHK(hipMemsetAsync(ctx->d_lookback, 0, lb_entries * 8, ctx->stream));
            const bool timed = kernel_ms && it >= warmup;
            if (timed) (void)hipEventRecord(evs[2 * (it - warmup)], ctx->stream);
            if (nt) pft_minimizer_launch(p->w, grid, ctx->stream, ka);
            if (timed) (void)hipEventRecord(evs[2 * (it - warmup) + 1], ctx->stream);
        }
        TCHK(hipGetLastError());
        TCHK(hipMemcpyAsync(ctx->d_ticket + 20, ctx->d_ticket, 4 * sizeof(u32), hipMemcpyDeviceToDevice, ctx->stream));
        if (kernel_ms && iters > 0) {
            TCHK(hipStreamSynchronize(ctx->stream));
            for (int i = 0; i < iters; ++i) (void)hipEventElapsedTime(&kernel_ms[i], evs[2 * i], evs[2 * i + 1]);
        }
        for (hipEvent_t e : evs)
            if (e) (void)hipEventDestroy(e);
        snprintf(tres->plan, sizeof tres->plan, "k_minimizer_pft<%d>", p->w);
        tres->plan_grid = grid;
        tres->plan_per_cu = per_cu;
        tres->n_tuples = 0;
    } else {
    ctx->defer = defer;
    rc = run_planned(ctx, tb, &p2, 0, &tres_slot, 0, 0, nullptr);
    ctx->defer = false;
    if (rc == BSK_OK && warmup + iters > 0) rc = run_planned_resizing(ctx, tb, &p2, 0, &tres_slot, warmup, iters, kernel_ms);
    tres = tres_slot;
    if (rc != BSK_OK) return done(rc);
    }
    lap("kernels (+sizing)");
    // 4. per-sequence flags
    TCHK(pool(7, (n ? n : 1) * 4, (void **)&sflags));
    TCHK(pool(8, (n ? n : 1) * 8, (void **)&sbad));
    TCHK(hipMemsetAsync(sflags, 0, (n ? n : 1) * 4, ctx->stream));
    TCHK(hipMemsetAsync(sbad, 0xff, (n ? n : 1) * 8, ctx->stream));
    if (nt) {
        hipLaunchKernelGGL(k_tile_flags, dim3(grid_for(ctx, nt, 256)), dim3(256), 0, ctx->stream, tres->status, tt.seq, tstart, nt, n, sflags,
                           sbad);
        TCHK(hipGetLastError());
    }
    // 5. the final, per-sequence result
    if (dense) {
        // (made before the tile kernel ran: it wrote into these arrays)
    } else if (old && !stream && old->hash && old->pos && old->alloc_cap >= tres->n_tuples + 64) {  // (stitched kinds: the old arrays are large enough)
        fin = old;
        old = nullptr;
    } else {
        drop_old();
        fin = new (std::nothrow) bsk_result();
        if (!fin) return done(BSK_ERR_NOMEM);
        fin->ctx = ctx;
        fin->n = n;
        fin->kind = p_in->kind;
        fin->has_pos = stream ? 0 : 1;
        TCHK(hipMalloc(&fin->status, n ? n : 1));
        TCHK(hipMalloc(&fin->wfirst, (n ? n : 1) * 8));
        TCHK(hipMalloc(&fin->wcount, (n ? n : 1) * 8));
    }
    if (dense) {
        // nothing to stitch: the tiles' owned tuples lie back to back in tile order
    } else if (two_strand) {  // twice the room; filled after k_tile_finish (k_two_strand)
        fin->cap = fin->alloc_cap = 2 * tres->cap + 64;
        TCHK(spare_take(ctx, (void **)&fin->hash, fin->cap * 8));
    } else if (stream) {  // the tile runs are adjacent: the tile result's value array IS the sequence result
        fin->hash = tres->hash;
        fin->cap = fin->alloc_cap = tres->cap;
        tres->hash = nullptr;
        tres->cap = tres->alloc_cap = 0;
        tres->main_cap = 0;
        tres->ovf_cap = 0;
    } else {
        u64 cap = tres->n_tuples + 64;  // the stitch keeps a subset of the tile tuples (deferred: n_tuples is the tile result's capacity)
        lap("flags");
        if (fin->hash) {
            cap = fin->alloc_cap;  // (the previous result's arrays)
        } else {
            TCHK(spare_take(ctx, (void **)&fin->hash, cap * 8));  // (a released result's arrays, if the context holds ones that fit: bsk_ctx::spare)
            TCHK(spare_take(ctx, (void **)&fin->pos, cap * 4));
            fin->cap = fin->alloc_cap = cap;
        }
        lap("result arrays");
        TCHK(pool(9, (nt + 1) * 8, (void **)&oexcl));
        TCHK(hipMemsetAsync(oexcl, 0, (nt + 1) * 8, ctx->stream));
        if (nt) {
            const u32 tunits = (u32)((nt + 63) / 64);
            rc = ensure_scratch(ctx, lb_words_with_heads(tunits), 0);
            if (rc != BSK_OK) return done(rc);
            TCHK(hipMemsetAsync(ctx->d_ticket, 0, 2 * sizeof(u32), ctx->stream));
            TCHK(hipMemsetAsync(ctx->d_lookback, 0, lb_words_with_heads(tunits) * 8, ctx->stream));
            StitchArgs sa;
            sa.nt = nt;
            sa.nunits = tunits;
            sa.trefs = tres->refs;
            sa.tcap = tres->alloc_cap ? tres->alloc_cap : tres->cap;
            sa.thash = tres->hash;
            sa.tpos = tres->pos;
            sa.shift = tt.shift;
            sa.keep = tt.keep;
            sa.oexcl = oexcl;
            sa.ohash = fin->hash;
            sa.opos = fin->pos;
            sa.cap = cap;
            sa.ticket = ctx->d_ticket;
            sa.lookback = ctx->d_lookback;
            hipLaunchKernelGGL(k_tile_stitch, dim3(std::min<u32>(tunits, (u32)ctx->cus * 32)), dim3(64), 0, ctx->stream, sa);  // latency-bound: every wave the CUs hold
            TCHK(hipGetLastError());
        }
    }
    lap("flags + stitch");
    TCHK(hipMemsetAsync(ctx->d_total, 0, 2 * sizeof(u64), ctx->stream));
    if (n) {
        hipLaunchKernelGGL(k_tile_finish, dim3(grid_for(ctx, n, 256)), dim3(256), 0, ctx->stream, seq, geo, tstart, (stream || dense) ? nullptr : oexcl,
                           tres->refs, prot ? nullptr : b->rflags, sflags, sbad, fin->wfirst, fin->wcount, fin->status, ctx->d_total, dense ? 1 : 0);
        TCHK(hipGetLastError());
    }
    if (two_strand && nt) {
        hipLaunchKernelGGL(k_two_strand, dim3((u32)std::min<u64>(nt, (u64)ctx->cus * 32)), dim3(256), 0, ctx->stream, tres->refs, tt.seq, nt,
                           tres->hash, fin->hash, fin->wfirst, fin->wcount, fin->status, p->k, b->n_nonacgt ? b->ascii : nullptr, b->aoff, b->pairs);
        TCHK(hipGetLastError());
        hipLaunchKernelGGL(k_two_strand_refs, dim3(grid_for(ctx, n, 256)), dim3(256), 0, ctx->stream, n, fin->wfirst, fin->wcount, fin->status,
                           ctx->d_total);
        TCHK(hipGetLastError());
    }
    if (async_final) {  // a class plan's tiled part: the totals stay on the device (k_adopt_wide reads wfirst / wcount), the flags wait in d_ticket[24]
        hipLaunchKernelGGL(k_tile_flag_word, dim3(1), dim3(1), 0, ctx->stream, ctx->d_ticket + 20, ctx->d_ticket, ctx->d_ticket + 24);
        TCHK(hipGetLastError());
        fin->n_tuples = fin->cap;  // (an upper bound: what the parent reserves and copies)
    } else {
        TCHK(hipMemcpyAsync(ctx->h_pinned, ctx->d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
        TCHK(hipMemcpyAsync(ctx->h_pinned + 2, ctx->d_ticket, 2 * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
        if (defer) TCHK(hipMemcpyAsync(ctx->h_pinned + 4, ctx->d_ticket + 20, 4 * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
        TCHK(hipStreamSynchronize(ctx->stream));
        const bool stitch_ovf = !stream && !dense && nt && ((u32 *)(ctx->h_pinned + 2))[1];
        if (defer && (stitch_ovf || ((u32 *)(ctx->h_pinned + 4))[1] || ((u32 *)(ctx->h_pinned + 4))[3] || (ctx->opt.test_overflow & 8u))) {
            // a slab, a list segment or an overflow region was too small for this batch: the old way sizes them by what the batch needs
            if (timing) fprintf(stderr, "[tiled] deferred launch overflowed (flags %u / %u, stitch %d): again with the sizing run\n", ((u32 *)(ctx->h_pinned + 4))[1], ((u32 *)(ctx->h_pinned + 4))[3], (int)stitch_ovf);
            (void)done(BSK_ERR_DEVICE);  // (releases `fin` and the tile batch)
            ctx->tile_sync = true;
            const int frc = sketch_tiled(ctx, b, p_in, circ_ext, result, warmup, iters, kernel_ms);
            ctx->tile_sync = false;
            return frc;
        }
        if (stitch_ovf) {
            ctx->err = "tile stitch overflow";
            return done(BSK_ERR_DEVICE);
        }
        fin->n_tuples = ctx->h_pinned[0];
    }
    snprintf(fin->plan, sizeof fin->plan, "%.70s (over tiles)", tres->plan);
    fin->plan_grid = tres->plan_grid;
    fin->plan_per_cu = tres->plan_per_cu;
#undef TCHK
    *result = fin;
    lap("finish");
    const int rcd = done(BSK_OK);
    lap("free temporaries");
    return rcd;
}

// the longest read the long packed syncmer plan takes (make_plan_enc's rule for k_syncmer_pkl, solved for the length): a pair of reads
// wants 1.12 x + margin rows of its column, x = 2 (1.5 windows / (k - s + 1) + 0.5)
static u32 syn_long_fit_bases(const bsk_ctx *ctx, const bsk_params *p) {
    const double x_max = ((double)pk_syncmer_pair_rows(true) - (double)ctx->opt.syn_margin) / 1.12;
    const double nwin_max = (x_max / 2.0 - 0.5) * (p->k - p->s + 1.0) / 1.5;
    const long long fit = (long long)(nwin_max + 1e-6) + 2LL * p->k - p->s - 2;
    return (u32)std::max<long long>(64, std::min<long long>(fit, (long long)pk_syncmer_max_bases(true)));
}
// from which sequence length a batch of this kind is cut into tiles (0: the kind does not tile)
u32 tile_min_for(const bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p) {
    const bool is_dna = b->alphabet == BSK_ALPHA_DNA;
    // (syncmers: beyond the packed kernels' reach the per-read 64-bit kernel falls to 140-160 Gbases/s of wall time -- its 28-tuple slabs
    // overflow -- and to 83 at 4 000 bases, where tiles run 170-210: scripts/dev/perf_midlen.py, round 4)
    // (round 5: from where the long packed plan's columns fill up -- 392 bases at k = 31, s = 11 -- not from a fixed 448: the reads in
    // between ran on k_syncmer_fast at 283 Gbases/s, tiles run them at ~420: scripts/dev/run_synlen.sh)
    // (off the tuned parameter points, scripts/dev/run_holes.sh: syncmers with k - s < 16 -- k=21 s=11, k=25 s=15 -- stayed on k_syncmer_fast
    // from 210 bases to the general threshold of 4 096: 250 -> 118 Gbases/s from 250 to 4 000 bases; minimizers with windows neither
    // packed kernel nor k_minimizer_dense takes (w >= 17) on k_minimizer_fast, whose 32-tuple columns overflow from ~300 bases:
    // w = 20: 812 at 250 bases, 392 / 294 / 192 at 400 / 700 / 4 000.  Both tile now from where their staged kernel stops fitting.)
    if (ctx->opt.tile_min) return ctx->opt.tile_min;
    if (is_dna && p->kind == BSK_SYNCMER && syn_long_plan_ok(ctx, p)) return std::min<u32>(kSynTileMin, syn_long_fit_bases(ctx, p));
    if (is_dna && p->kind == BSK_MINIMIZER && !pkd_minimizer_supported(p->w) && !dense_minimizer_supported(p->w) && fast_minimizer_supported(p->w) && !ctx->opt.force_generic)
        return std::min<u32>(4096u, (u32)(11 * (p->w + 1) + p->k + p->w));  // 22 expected tuples of the 32 a lane stages
    return (!is_dna || kind_has_pos(p->kind)) ? 4096u : 16u * (BSK_NT_FAST_WORDS - 2);
}

