// biosketch.hip -- C ABI (include/biosketch.h) of the MI355X k-mer sketching engine.
// Host side: contexts, batches (device-resident 2-bit packed reads), results (device-resident CSR tuples), translation and the bsk_sketch
// entry points.  The planner is planner.hip, the launches launch.hip, long sequences tiles.hip, class plans classes.hip (host_internal.hpp
// is what they share).  gfx950 only; there is no CPU implementation behind this ABI: without a device every compute entry fails.
#include "host_internal.hpp"
#include "kernels_host.hpp"

// context
// ------------------------------------------------------------------------------------
extern "C" int bsk_abi_version(void) { return BSK_ABI_VERSION; }

extern "C" const char *bsk_err_name(int e) {
    switch (e) {
        case BSK_OK: return "ok";
        case BSK_ERR_INVALID_K: return "ErrInvalidK";
        case BSK_ERR_EMPTY_SEQ: return "ErrEmptySeq";
        case BSK_ERR_SHORT_SEQ: return "ErrShortSeq";
        case BSK_ERR_ILLEGAL_BASE: return "ErrIllegalBase";
        case BSK_ERR_K_TOO_LARGE: return "ErrKTooLarge";
        case BSK_ERR_INVALID_M: return "ErrInvalidM";
        case BSK_ERR_INVALID_SCALE: return "ErrInvalidScale";
        case BSK_ERR_INVALID_S: return "ErrInvalidS";
        case BSK_ERR_INVALID_W: return "ErrInvalidW";
        case BSK_ERR_BUF_NIL: return "ErrBufNil";
        case BSK_ERR_BUF_NOT_EMPTY: return "ErrBufNotEmpty";
        case BSK_ERR_ARG: return "bad argument";
        case BSK_ERR_NOMEM: return "out of memory";
        case BSK_ERR_DEVICE: return "device error";
        case BSK_ERR_UNSUPPORTED: return "unsupported";
        case BSK_ERR_NO_DEVICE: return "no gfx950 device";
        case BSK_ERR_IO: return "fastx: cannot open or read the file";
        case BSK_ERR_NOT_FASTX: return "fastx: invalid FASTA/Q format";
        case BSK_ERR_BAD_FASTQ: return "fastx: bad FASTQ format";
        case BSK_ERR_STOPPED: return "pipeline: stopped by the consumer";
        default: return "unknown";
    }
}

extern "C" int bsk_device_count(int *n) {
    if (!n) return BSK_ERR_ARG;
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c = 0;
    }
    *n = c;
    return BSK_OK;
}

extern "C" int bsk_ctx_create(int device, bsk_ctx **out) {
    if (!out) return BSK_ERR_ARG;
    *out = nullptr;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c <= 0) {
        (void)hipGetLastError();
        return BSK_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= c) return BSK_ERR_ARG;
    bsk_ctx *ctx = new (std::nothrow) bsk_ctx();
    if (!ctx) return BSK_ERR_NOMEM;
    ctx->device = device;
    hipDeviceProp_t prop;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) {
        delete ctx;
        return BSK_ERR_DEVICE;
    }
    ctx->cus = prop.multiProcessorCount;
    ctx->opt.load();  // the developer switches: once per context
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc(&ctx->d_ticket, 32 * sizeof(u32)) != hipSuccess || hipMalloc(&ctx->d_total, 8 * sizeof(u64)) != hipSuccess ||
        hipHostMalloc(&ctx->h_pinned, 8 * sizeof(u64)) != hipSuccess) {
        bsk_ctx_destroy(ctx);
        return BSK_ERR_DEVICE;
    }
    spare_register(ctx, true);
    *out = ctx;
    return BSK_OK;
}

extern "C" void bsk_ctx_destroy(bsk_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    spare_register(ctx, false);
    spare_flush(ctx);
    if (ctx->side) {
        bsk_ctx_destroy(ctx->side);
        ctx->side = nullptr;
    }
    if (ctx->ev_side_done) (void)hipEventDestroy(ctx->ev_side_done);
    if (ctx->ev_adopted) (void)hipEventDestroy(ctx->ev_adopted);
    if (ctx->ev_tiled) (void)hipEventDestroy(ctx->ev_tiled);
    if (ctx->ev_mix0) (void)hipEventDestroy(ctx->ev_mix0);
    if (ctx->ev_mix1) (void)hipEventDestroy(ctx->ev_mix1);
    bsk_comm_destroy(ctx);
    (void)hipFree(ctx->d_ticket);
    (void)hipFree(ctx->d_total);
    (void)hipFree(ctx->d_lookback);
    (void)hipFree(ctx->d_ring_h);
    (void)hipFree(ctx->d_ring_p);
    (void)hipFree(ctx->d_lut);
    for (void *t : ctx->tmp) (void)hipFree(t);
    if (ctx->tile_res) bsk_result_release(ctx->tile_res);
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    if (ctx->h_refs) (void)hipHostFree(ctx->h_refs);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int bsk_ctx_sync(bsk_ctx *ctx) {
    if (!ctx) return BSK_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return BSK_OK;
}

extern "C" const char *bsk_last_error(const bsk_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

// ------------------------------------------------------------------------------------
// batches
// ------------------------------------------------------------------------------------
int grid_for(bsk_ctx *ctx, u64 items, int block) {
    u64 g = (items + block - 1) / block;
    u64 cap = (u64)ctx->cus * 16;
    return (int)std::max<u64>(1, std::min(g, cap));
}

extern "C" void bsk_batch_destroy(bsk_batch *b) {
    if (!b) return;
    if (b->ctx) (void)hipSetDevice(b->ctx->device);
    delete b->hist;
    delete b->odd;
    if (b->borrowed) {  // a view of another batch (class plans): its descriptors live in the context's pool, only the binned copies are its own
        (void)hipFree(b->bdesc);
        (void)hipFree(b->bflags);
        delete b;
        return;
    }
    (void)hipFree(b->d_odd);
    if (!b->alias) {
        (void)hipFree(b->words);
        (void)hipFree(b->ascii);
    }
    (void)hipFree(b->desc);
    (void)hipFree(b->fw);
    (void)hipFree(b->llen);
    (void)hipFree(b->adesc);
    (void)hipFree(b->subset);
    (void)hipFree(b->wbits);
    (void)hipFree(b->rflags);
    (void)hipFree(b->bdesc);
    (void)hipFree(b->bflags);
    (void)hipFree(b->aoff);
    (void)hipFree(b->spare_ascii);
    (void)hipFree(b->spare_aoff);
    delete b;
}

// the sequences outside the histogram's fullest bucket, if they are few (bsk_batch::odd); len(r) = bases of sequence r
// The length-binned view of a ragged batch of short reads, built WITH the batch (its lengths are known there, the pass runs behind the
// pack kernel on the same stream): classes so fine -- (longest - shortest) / 63 bases, one or two bases for trimmed reads -- that the
// reads of a unit end within a base or two of each other whatever the plan's block of w k-mers or k - s s-mers is, so that no plan
// needs a pass of its own (round 4: k_bin_desc per plan was 10 % of the minimizer kernel on a fresh batch; bsk_batch_prepare now
// reports 0 for such a batch).  Only batches the planner can bin at all (bin_gran_for); BSK_NO_BIN_EARLY=1: per plan, as before.
int bin_with_batch(bsk_ctx *ctx, bsk_batch *b) {
    b->bin_gran = 0;
    b->bin_early = false;
    if (ctx->opt.no_bin || ctx->opt.no_bin_early || b->alphabet != BSK_ALPHA_DNA || b->uniform_len || !b->desc || b->alias || b->maxlen >= 4096u || b->n < (u64)ctx->opt.bin_min || !b->hist) return BSK_OK;
    u32 shortest = b->maxlen;
    for (int i = 0; i < LenHist::NB; ++i)
        if (b->hist->cnt[i] && b->hist->lo[i] < shortest) shortest = b->hist->lo[i];
    const u32 lo = shortest ? shortest - 1 : 0;
    const u32 gran = std::max<u32>(1u, (b->maxlen - lo + 125u) / 126u);  // classes 1 .. 126: the reads of a chunk in order of length when they span 126 bases or fewer
    const int rc = ensure_binned(ctx, b, lo, gran, 0, 0, 0, true);
    if (rc == BSK_OK) b->bin_early = true;
    return rc;
}

template <class LenOf>
static void collect_odd(bsk_batch *b, u64 n, LenOf len) {
    delete b->odd;
    b->odd = nullptr;
    b->modal_bucket = -1;
    if (!b->hist || n >= (1ULL << 32)) return;
    int mb = 0;
    for (int i = 1; i < LenHist::NB; ++i)
        if (b->hist->cnt[i] > b->hist->cnt[mb]) mb = i;
    const u64 others = n - b->hist->cnt[mb];
    if (others == 0 || others > n / 20) return;
    auto *v = new (std::nothrow) std::vector<u64>();
    if (!v) return;
    v->reserve((size_t)others);
    for (u64 r = 0; r < n; ++r) {
        const u64 L = len(r);
        if (LenHist::bucket(L) != mb) v->push_back((r << 32) | L);
    }
    b->odd = v;
    b->modal_bucket = mb;
}
// the list on the device too (class plans split long lists there: k_odd_split); with the batch's other uploads, on its stream
static void upload_odd(bsk_ctx *ctx, bsk_batch *b) {
    (void)hipFree(b->d_odd);
    b->d_odd = nullptr;
    if (!b->odd || b->odd->size() < 65536) return;
    if (hipMalloc(&b->d_odd, b->odd->size() * sizeof(u64)) != hipSuccess) {
        (void)hipGetLastError();
        b->d_odd = nullptr;
        return;
    }
    if (hipMemcpyAsync(b->d_odd, b->odd->data(), b->odd->size() * sizeof(u64), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(b->d_odd);
        b->d_odd = nullptr;
    }
}

// Slack behind the packed words.  Every prefetching kernel loads a fixed number of words from every read's FIRST word -- also the last
// read's, a zero-length read's and a tile's that aliases the end of words[]: k_minimizer_pk / k_minimizer_ring / k_syncmer_pk 16 words,
// k_syncmer_pkl BSK_SYNPKL_NW = 32 (register path, no LDS-DMA), DnaResidues::issue4 maxlen/16 + 10.  The pad is the WIDEST prefetch + 1,
// whatever the batch's longest read (the long syncmer plan is reachable with a small maxlen: k-s = 21..24, dense selections).
// The smallest read length any kind tiles from (sketch_impl's tile_min: syncmers on the long packed plan from 448 bases, stream kinds from
// 16 (BSK_NT_FAST_WORDS - 2) = 512): batch creation keeps the non-ACGT word bits of every batch that MAY be tiled.
u64 pad_words(u32 maxlen) { return (u64)maxlen / 16 + kMaxPrefetchWords + 1; }
u32 env_u32(const char *name, u32 dflt) {
    const char *v = getenv(name);
    return v && *v ? (u32)strtoul(v, nullptr, 10) : dflt;
}
u32 min_tile_min(const bsk_ctx *ctx) { return ctx->opt.tile_min ? ctx->opt.tile_min : std::min<u32>(64u, 16u * (BSK_NT_FAST_WORDS - 2)); }  // (syncmers and wide-window minimizers tile from where their staged kernel stops fitting: tile_min_for)
void BskOpts::load() {
    auto on = [](const char *n) { return getenv(n) != nullptr; };
    force_generic = on("BSK_FORCE_GENERIC");
    no_mixed = on("BSK_NO_MIXED");
    no_dense = on("BSK_NO_DENSE");
    no_pk = on("BSK_NO_PK");
    no_ring = on("BSK_NO_RING");
    no_pkd = on("BSK_NO_PKD");
    no_side_early = on("BSK_NO_SIDE_EARLY");
    no_side_dense = on("BSK_NO_SIDE_DENSE");
    no_bin = on("BSK_NO_BIN");
    no_bin_early = on("BSK_NO_BIN_EARLY");
    compact = on("BSK_COMPACT");
    ring = on("BSK_RING");
    ring_max = env_u32("BSK_RING_MAX", 0);
    ring_sel10 = env_u32("BSK_RING_SEL10", 0);
    bin_min = env_u32("BSK_BIN_MIN", 1024);
    no_class = on("BSK_NO_CLASS");
    syn_sel = on("BSK_SYN_SEL");
    class_min = env_u32("BSK_CLASS_MIN", 16384);
    class_force = on("BSK_CLASS_FORCE");
    class_view = on("BSK_CLASS_VIEW");
    no_syn_pf = on("BSK_NO_SYN_PF");
    pf_density = env_u32("BSK_PF_DENSITY", 0);
    no_syn_long = on("BSK_NO_SYN_LONG");  // dev: reads beyond k_syncmer_pk's limits go to k_syncmer_fast as before round 4
    syn_margin = (int)env_u32("BSK_SYN_MARGIN", (u32)PlannerTable::syn_margin_rows + 64) - 64;  // dev: rows of slack the planner wants in k_syncmer_pk's columns (BSK_SYN_MARGIN = 64 + margin)
    no_tiles = on("BSK_NO_TILES");
    no_spare = on("BSK_NO_SPARE");  // released results' arrays are freed, not kept for the next result (bsk_ctx::spare)
    no_tile_cache = on("BSK_NO_TILE_CACHE");
    tile_dense = on("BSK_TILE_DENSE");  // minimizers of long sequences by the dense tile kernel (k_minimizer_pft: final tuples, no stitch pass).  Exact, and measured no faster than slabs + k_tile_stitch (6.3 against 5.9 ms for 2 10^9 bases -- its from-scratch emit is VALU the stitch pass pays in HBM time): opt-in
    no_tile_defer = on("BSK_NO_TILE_DEFER");  // dev: tiled calls with the host round trips of rounds 2-5 (tile count, sizing run, totals)
    no_group_gather = on("BSK_NO_GROUP_GATHER");  // dev: bsk_result_compact / _fetch_narrow with one (part of a) wavefront per sequence, as before round 4
    timing = on("BSK_TIMING");
    no_fused_translate = on("BSK_NO_FUSED_TRANSLATE");
    sets_no_small = on("BSK_SETS_NO_SMALL");
    wpr = env_u32("BSK_WPR", 0);
    seg = env_u32("BSK_SEG", 0);
    dense_min = env_u32("BSK_DENSE_MIN", (u32)PlannerTable::dense_min);
    waves_per_cu = env_u32("BSK_WAVES_PER_CU", 0);
    tile_min = env_u32("BSK_TILE_MIN", 0);
    tile_pos = env_u32("BSK_TILE_POS", 0);
    test_overflow = env_u32("BSK_TEST_OVERFLOW", 0);
}
extern "C" int bsk_build_has_experiments(void) {
#ifdef BSK_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}
extern "C" int bsk_ctx_reload_options(bsk_ctx *ctx) {
    if (!ctx) return BSK_ERR_ARG;
    ctx->opt.load();
    if (ctx->side) ctx->side->opt = ctx->opt;
    return BSK_OK;
}
// the side context of a context's class plans (stream + scratch of its own), and the two events that order its stream with the main one
bsk_ctx *side_ctx(bsk_ctx *ctx) {
    if (!ctx->side) {
        bsk_ctx *s = nullptr;
        if (bsk_ctx_create(ctx->device, &s) != BSK_OK) return nullptr;
        if (hipEventCreateWithFlags(&ctx->ev_side_done, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_adopted, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->ev_mix0, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_mix1, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->ev_tiled, hipEventDisableTiming) != hipSuccess) {
            bsk_ctx_destroy(s);
            return nullptr;
        }
        ctx->side = s;
    }
    ctx->side->opt = ctx->opt;
    return ctx->side;
}

// donor: a batch whose device buffers may be taken over (bsk_batch_refill_ascii); it is consumed
static int batch_from_ascii_impl(bsk_ctx *ctx, const uint8_t *bytes, const uint64_t *offsets, uint64_t n, int alphabet, bsk_batch *donor,
                                 bsk_batch **out) {
    if (!ctx || !out || (!offsets && n) || (n && !bytes && offsets[n] > 0)) return fail_arg(ctx, "bsk_batch_from_ascii: null argument");
    if (alphabet < BSK_ALPHA_DNA || alphabet > BSK_ALPHA_UNLIMIT) return fail_arg(ctx, "bad alphabet");
    const int pairs = alphabet == BSK_ALPHA_PROTEIN ? BSK_ALPHA_DNA : alphabet;  // which PairLetter the two-strand k-mer mode uses
    if (alphabet != BSK_ALPHA_PROTEIN) alphabet = BSK_ALPHA_DNA;                 // nucleotides: one engine alphabet
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (donor && !donor->ascii && donor->spare_ascii) {  // a pure-ACGT donor parked its ASCII buffers
        donor->ascii = donor->spare_ascii;
        donor->aoff = donor->spare_aoff;
        donor->spare_ascii = nullptr;
        donor->spare_aoff = nullptr;
    }
    // device buffer of at least `bytes`: the donor's if it is large enough, else a fresh one with 1/8 of slack
    auto take = [&](void **dst, size_t *cap_dst, size_t need, void **src, size_t *cap_src) -> hipError_t {
        if (donor && src && *src && *cap_src >= need) {
            *dst = *src;
            *cap_dst = *cap_src;
            *src = nullptr;
            *cap_src = 0;
            return hipSuccess;
        }
        const size_t want = donor ? need + need / 8 + 256 : need;
        const hipError_t e = hipMalloc(dst, want ? want : 1);
        if (e == hipSuccess) *cap_dst = want;
        return e;
    };
    struct DonorGuard {  // whatever was not taken over goes away with the donor
        bsk_batch *d;
        ~DonorGuard() { if (d) bsk_batch_destroy(d); }
    } donor_guard{donor};
    bsk_batch *b = new (std::nothrow) bsk_batch();
    if (!b) return BSK_ERR_NOMEM;
    b->ctx = ctx;
    b->alphabet = alphabet;
    b->pairs = pairs;
    b->n = n;
    if (donor) {  // the buffers of the donor's length-binned view (its contents are the donor's: bin_gran stays 0)
        std::swap(b->bdesc, donor->bdesc);
        std::swap(b->c_bdesc, donor->c_bdesc);
        std::swap(b->bflags, donor->bflags);
        std::swap(b->c_bflags, donor->c_bflags);
    }
    const u64 nbytes = n ? offsets[n] : 0;
    b->n_bases = nbytes;
    u32 maxlen = 0;
    bool uniform = true;
    for (u64 r = 0; r < n; ++r) {
        if (offsets[r + 1] < offsets[r]) {
            delete b;
            return fail_arg(ctx, "offsets not monotone");
        }
        u64 L = offsets[r + 1] - offsets[r];
        if (L >= (1ULL << 31) || (L >= (1ULL << 24) && alphabet != BSK_ALPHA_DNA)) {
            delete b;
            ctx->err = "sequence too long (DNA: 2^31 bases, positions carry the strand in bit 31; protein: 2^24 residues)";
            return BSK_ERR_UNSUPPORTED;
        }
        maxlen = std::max<u32>(maxlen, (u32)L);
        if (L != offsets[1] - offsets[0]) uniform = false;
    }
    b->maxlen = maxlen;
    b->uniform_len = (n && uniform) ? maxlen : 0;  // fixed-length batches (most FASTQ files): closed-form output offsets, no per-lane window test
    int rc = BSK_OK;
    auto bail = [&](int code) {
        bsk_batch_destroy(b);
        return code;
    };
#define BCHK(call)                                                  \
    do {                                                            \
        hipError_t e__ = (call);                                    \
        if (e__ != hipSuccess) return bail(fail_hip(ctx, e__, #call)); \
    } while (0)
    // ascii + offsets to the device
    BCHK(take((void **)&b->ascii, &b->c_ascii, nbytes + BSK_ASCII_PAD, donor ? (void **)&donor->ascii : nullptr, donor ? &donor->c_ascii : nullptr));
    BCHK(take((void **)&b->aoff, &b->c_aoff, (n + 1) * sizeof(u64), donor ? (void **)&donor->aoff : nullptr, donor ? &donor->c_aoff : nullptr));
    if (nbytes) BCHK(hipMemcpyAsync(b->ascii, bytes, nbytes, hipMemcpyHostToDevice, ctx->stream));
    if (n) BCHK(hipMemcpyAsync(b->aoff, offsets, (n + 1) * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
    else {
        u64 z = 0;
        BCHK(hipMemcpyAsync(b->aoff, &z, sizeof z, hipMemcpyHostToDevice, ctx->stream));
    }
    b->device_bytes = nbytes + 64 + (n + 1) * 8;
    if (alphabet == BSK_ALPHA_DNA) {
        const bool wide = maxlen >= (1u << 24);  // desc cannot hold such a length: sequences are located by fw + llen
        // (descriptors are built in the context's pinned staging words: a pageable std::vector made the 2 MB copy of every streamed
        // chunk a staged, synchronous one)
        if (ctx->h_refs_cap < n + 1) {
            if (ctx->h_refs) (void)hipHostFree(ctx->h_refs);
            ctx->h_refs = nullptr;
            ctx->h_refs_cap = 0;
            const size_t want = (n + 1) + (n + 1) / 4 + 64;
            BCHK(hipHostMalloc(&ctx->h_refs, want * 8));
            ctx->h_refs_cap = want;
        }
        u64 *const desc = ctx->h_refs;
        std::vector<u64> llen(wide ? n : 0);
        u64 w = 0;
        LenHist *hist = (!uniform && !wide && n) ? new (std::nothrow) LenHist() : nullptr;  // what the class plans are cut from (run_classed)
        for (u64 r = 0; r < n; ++r) {
            u64 L = offsets[r + 1] - offsets[r];
            desc[r] = wide ? w : ((w << 24) | L);
            if (wide) llen[r] = L;
            if (hist) hist->add(L);
            w += (L + 15) / 16;
        }
        delete b->hist;
        b->hist = hist;
        collect_odd(b, n, [&](u64 r) { return offsets[r + 1] - offsets[r]; });
        upload_odd(ctx, b);
        b->n_words = w;
        const u64 alloc_words = w + pad_words(maxlen);
        BCHK(take((void **)&b->words, &b->c_words, alloc_words * sizeof(u32), donor ? (void **)&donor->words : nullptr, donor ? &donor->c_words : nullptr));
        if (wide) BCHK(hipMalloc(&b->fw, (n ? n : 1) * sizeof(u64)));
        else BCHK(take((void **)&b->desc, &b->c_desc, (n ? n : 1) * sizeof(u64), donor ? (void **)&donor->desc : nullptr, donor ? &donor->c_desc : nullptr));
        if (wide) {
            BCHK(hipMalloc(&b->llen, n * sizeof(u64)));
            BCHK(hipMemcpyAsync(b->llen, llen.data(), n * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
        }
        BCHK(take((void **)&b->rflags, &b->c_rflags, n ? n : 1, donor ? (void **)&donor->rflags : nullptr, donor ? &donor->c_rflags : nullptr));
        BCHK(hipMemsetAsync(b->words, 0, alloc_words * sizeof(u32), ctx->stream));
        BCHK(hipMemsetAsync(b->rflags, 0, n ? n : 1, ctx->stream));
        if (n) BCHK(hipMemcpyAsync(wide ? b->fw : b->desc, desc, n * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
        BCHK(hipMemsetAsync(ctx->d_ticket, 0, 2 * sizeof(u32), ctx->stream));
        if (maxlen > min_tile_min(ctx) && w) {  // this batch may be tiled: remember which words hold non-ACGT letters
            BCHK(hipMalloc(&b->wbits, ((w + 31) / 32) * sizeof(u32)));
            BCHK(hipMemsetAsync(b->wbits, 0, ((w + 31) / 32) * sizeof(u32), ctx->stream));
        }
        if (n && w) {
            hipLaunchKernelGGL(k_pack, dim3(grid_for(ctx, w, 256)), dim3(256), 0, ctx->stream, b->ascii, b->aoff, b->desc, b->fw, n, w,
                               b->words, b->rflags, ctx->d_ticket, b->wbits);
            hipLaunchKernelGGL(k_count_flags, dim3(grid_for(ctx, n, 256)), dim3(256), 0, ctx->stream, b->rflags, n,
                               ctx->d_ticket + 1);
        }
        if (!wide && (rc = bin_with_batch(ctx, b)) != BSK_OK) return bail(rc);  // (ragged short reads: the length-binned view, behind the pack kernel)
        BCHK(hipMemcpyAsync(ctx->h_pinned, ctx->d_ticket, 2 * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
        BCHK(hipStreamSynchronize(ctx->stream));  // also: desc (host vector) no longer needed after this
        BCHK(hipGetLastError());
        b->n_nonacgt = ((u32 *)ctx->h_pinned)[1];
        b->device_bytes += alloc_words * 4 + n * 9;
        if (b->n_nonacgt == 0) {  // pure ACGT: the 2-bit stream is all the kernels need
            if (donor) {  // streaming caller: park the buffers for the next refill instead of freeing them
                b->spare_ascii = b->ascii;
                b->spare_aoff = b->aoff;
            } else {
                (void)hipFree(b->ascii);
                (void)hipFree(b->aoff);
                b->c_ascii = b->c_aoff = 0;
            }
            b->ascii = nullptr;
            b->aoff = nullptr;
            b->device_bytes -= nbytes + 64 + (n + 1) * 8;
        } else if ((rc = build_subset(ctx, b)) != BSK_OK) {
            return bail(rc);
        }
    } else {
        BCHK(hipStreamSynchronize(ctx->stream));
    }
#undef BCHK
    *out = b;
    return rc;
}

extern "C" int bsk_batch_from_ascii(bsk_ctx *ctx, const uint8_t *bytes, const uint64_t *offsets, uint64_t n, int alphabet,
                                    bsk_batch **out) {
    return batch_from_ascii_impl(ctx, bytes, offsets, n, alphabet, nullptr, out);
}

extern "C" int bsk_batch_refill_ascii(bsk_ctx *ctx, bsk_batch **batch, const uint8_t *bytes, const uint64_t *offsets, uint64_t n, int alphabet) {
    if (!ctx || !batch) return fail_arg(ctx, "bsk_batch_refill_ascii: null argument");
    bsk_batch *old = *batch;
    if (old && (old->ctx != ctx || old->alias)) return fail_arg(ctx, "bsk_batch_refill_ascii: the batch belongs to another context");
    *batch = nullptr;  // consumed whatever happens
    return batch_from_ascii_impl(ctx, bytes, offsets, n, alphabet, old, batch);
}

// donor: a batch whose device buffers may be taken over (bsk_batch_refill_packed); it is consumed
static int batch_from_packed_impl(bsk_ctx *ctx, const uint32_t *words, uint64_t n_words, const uint64_t *desc, uint64_t n, bsk_batch *donor,
                                  bsk_batch **out) {
    struct DonorGuard {
        bsk_batch *d;
        ~DonorGuard() { if (d) bsk_batch_destroy(d); }
    } donor_guard{donor};
    if (!ctx || !out || (n && !desc) || (n_words && !words)) return fail_arg(ctx, "bsk_batch_from_packed: null argument");
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    u32 maxlen = 0;
    u64 nb = 0;
    bool uniform = true;
    std::unique_ptr<LenHist> hist;
    for (u64 r = 0; r < n; ++r) {
        u64 L = desc[r] & 0xffffffULL, w0 = desc[r] >> 24;
        if (w0 + (L + 15) / 16 > n_words) return fail_arg(ctx, "desc points outside words[]");
        maxlen = std::max<u32>(maxlen, (u32)L);
        nb += L;
        if (L != (desc[0] & 0xffffffULL) && uniform) {  // the first read of another length: the histogram starts here (run_classed)
            uniform = false;
            hist.reset(new (std::nothrow) LenHist());
            if (hist) {
                hist->cnt[LenHist::bucket(desc[0] & 0xffffffULL)] = r;
                hist->bases[LenHist::bucket(desc[0] & 0xffffffULL)] = r * (desc[0] & 0xffffffULL);
                hist->hi[LenHist::bucket(desc[0] & 0xffffffULL)] = (u32)(desc[0] & 0xffffffULL);
                hist->lo[LenHist::bucket(desc[0] & 0xffffffULL)] = (u32)(desc[0] & 0xffffffULL);
            }
        }
        if (hist) hist->add(L);
    }
    bsk_batch *b = new (std::nothrow) bsk_batch();
    if (!b) return BSK_ERR_NOMEM;
    b->hist = hist.release();
    collect_odd(b, n, [&](u64 r) { return desc[r] & 0xffffffULL; });
    upload_odd(ctx, b);
    b->ctx = ctx;
    b->alphabet = BSK_ALPHA_DNA;
    b->n = n;
    b->n_bases = nb;
    b->n_words = n_words;
    b->maxlen = maxlen;
    b->uniform_len = (n && uniform) ? maxlen : 0;
    const u64 alloc_words = n_words + pad_words(maxlen);
    // device buffer of at least `need` bytes: the donor's if it is large enough (a streaming caller refills one batch object per
    // stream: hipFree / hipMalloc per chunk would synchronise the device), else a fresh one with 1/8 of slack
    auto take = [&](void **dst, size_t *cap_dst, size_t need, void **src, size_t *cap_src) -> hipError_t {
        if (donor && *src && *cap_src >= need) {
            *dst = *src;
            *cap_dst = *cap_src;
            *src = nullptr;
            *cap_src = 0;
            return hipSuccess;
        }
        const size_t want = donor ? need + need / 8 + 256 : need;
        const hipError_t e2 = hipMalloc(dst, want ? want : 1);
        if (e2 == hipSuccess) *cap_dst = want;
        return e2;
    };
    if (donor) {
        std::swap(b->bdesc, donor->bdesc);
        std::swap(b->c_bdesc, donor->c_bdesc);
        std::swap(b->bflags, donor->bflags);
        std::swap(b->c_bflags, donor->c_bflags);
        // (an ASCII-refilled donor parks its ASCII buffers: they stay with the batch object for a later ASCII refill)
        std::swap(b->spare_ascii, donor->spare_ascii);
        std::swap(b->spare_aoff, donor->spare_aoff);
        if (donor->ascii && !b->spare_ascii) {
            b->spare_ascii = donor->ascii;
            b->spare_aoff = donor->aoff;
            donor->ascii = nullptr;
            donor->aoff = nullptr;
        }
        b->c_ascii = donor->c_ascii;
        b->c_aoff = donor->c_aoff;
    }
    hipError_t e;
    if ((e = take((void **)&b->words, &b->c_words, alloc_words * 4, donor ? (void **)&donor->words : nullptr, donor ? &donor->c_words : nullptr)) != hipSuccess ||
        (e = take((void **)&b->desc, &b->c_desc, (n ? n : 1) * 8, donor ? (void **)&donor->desc : nullptr, donor ? &donor->c_desc : nullptr)) != hipSuccess ||
        (e = take((void **)&b->rflags, &b->c_rflags, n ? n : 1, donor ? (void **)&donor->rflags : nullptr, donor ? &donor->c_rflags : nullptr)) != hipSuccess ||
        (e = hipMemsetAsync(b->words + n_words, 0, (alloc_words - n_words) * 4, ctx->stream)) != hipSuccess ||
        (e = hipMemsetAsync(b->rflags, 0, n ? n : 1, ctx->stream)) != hipSuccess ||
        (n_words && (e = hipMemcpyAsync(b->words, words, n_words * 4, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) ||
        (n && (e = hipMemcpyAsync(b->desc, desc, n * 8, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) ||
        (e = hipStreamSynchronize(ctx->stream)) != hipSuccess) {
        bsk_batch_destroy(b);
        return fail_hip(ctx, e, "bsk_batch_from_packed");
    }
    b->device_bytes = alloc_words * 4 + n * 9;
    {
        const int brc = bin_with_batch(ctx, b);  // (ragged short reads: the length-binned view comes with the batch)
        if (brc != BSK_OK) {
            bsk_batch_destroy(b);
            return brc;
        }
    }
    *out = b;
    return BSK_OK;
}

extern "C" int bsk_batch_from_packed(bsk_ctx *ctx, const uint32_t *words, uint64_t n_words, const uint64_t *desc, uint64_t n,
                                     bsk_batch **out) {
    return batch_from_packed_impl(ctx, words, n_words, desc, n, nullptr, out);
}

extern "C" int bsk_batch_refill_packed(bsk_ctx *ctx, bsk_batch **batch, const uint32_t *words, uint64_t n_words, const uint64_t *desc, uint64_t n) {
    if (!ctx || !batch) return fail_arg(ctx, "bsk_batch_refill_packed: null argument");
    bsk_batch *old = *batch;
    if (old && (old->ctx != ctx || old->alias)) return fail_arg(ctx, "bsk_batch_refill_packed: the batch belongs to another context");
    *batch = nullptr;  // consumed whatever happens
    return batch_from_packed_impl(ctx, words, n_words, desc, n, old, batch);
}

extern "C" int bsk_batch_synth(bsk_ctx *ctx, int alphabet, uint64_t n, uint32_t len, uint64_t seed, bsk_batch **out) {
    if (!ctx || !out) return fail_arg(ctx, "bsk_batch_synth: null argument");
    if (len == 0 || len >= (1u << 24) || n == 0) return fail_arg(ctx, "bsk_batch_synth: bad n/len");
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    bsk_batch *b = new (std::nothrow) bsk_batch();
    if (!b) return BSK_ERR_NOMEM;
    b->ctx = ctx;
    b->alphabet = alphabet;
    b->n = n;
    b->n_bases = n * len;
    b->maxlen = len;
    b->uniform_len = len;
    hipError_t e = hipSuccess;
    if (alphabet == BSK_ALPHA_DNA) {
        const u32 wpr = (len + 15) / 16;
        b->n_words = n * wpr;
        const u64 alloc_words = b->n_words + pad_words(len);
        if ((e = hipMalloc(&b->words, alloc_words * 4)) == hipSuccess && (e = hipMalloc(&b->desc, n * 8)) == hipSuccess &&
            (e = hipMalloc(&b->rflags, n)) == hipSuccess &&
            (e = hipMemsetAsync(b->words + b->n_words, 0, pad_words(len) * 4, ctx->stream)) == hipSuccess) {
            hipLaunchKernelGGL(k_synth_dna, dim3(grid_for(ctx, b->n_words, 256)), dim3(256), 0, ctx->stream, b->words, b->desc,
                               b->rflags, n, len, wpr, seed);
            e = hipGetLastError();
        }
        b->device_bytes = alloc_words * 4 + n * 9;
    } else if (alphabet == BSK_ALPHA_PROTEIN) {
        if ((e = hipMalloc(&b->ascii, n * len + BSK_ASCII_PAD)) == hipSuccess && (e = hipMalloc(&b->aoff, (n + 1) * 8)) == hipSuccess) {
            hipLaunchKernelGGL(k_synth_protein, dim3(grid_for(ctx, n * len, 256)), dim3(256), 0, ctx->stream, b->ascii, b->aoff, n,
                               len, seed);
            e = hipGetLastError();
        }
        b->device_bytes = n * len + 64 + (n + 1) * 8;
    } else {
        delete b;
        return fail_arg(ctx, "bad alphabet");
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        bsk_batch_destroy(b);
        return fail_hip(ctx, e, "bsk_batch_synth");
    }
    *out = b;
    return BSK_OK;
}

extern "C" int bsk_batch_info(const bsk_batch *b, uint64_t *n_reads, uint64_t *n_bases, uint64_t *device_bytes,
                              uint64_t *n_non_acgt_reads) {
    if (!b) return BSK_ERR_ARG;
    if (n_reads) *n_reads = b->n;
    if (n_bases) *n_bases = b->n_bases;
    if (device_bytes) *device_bytes = b->device_bytes;
    if (n_non_acgt_reads) *n_non_acgt_reads = b->n_nonacgt;
    return BSK_OK;
}

extern "C" int bsk_batch_fetch_ascii(bsk_ctx *ctx, const bsk_batch *b, uint64_t first, uint64_t count, uint8_t *bytes,
                                     uint64_t bytes_cap, uint64_t *offsets) {
    if (!ctx || !b || !offsets || (!bytes && bytes_cap)) return fail_arg(ctx, "bsk_batch_fetch_ascii: null argument");
    if (first + count > b->n) return fail_arg(ctx, "bsk_batch_fetch_ascii: range outside batch");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    offsets[0] = 0;
    if (count == 0) return BSK_OK;
    if (b->ascii) {  // exact original bytes
        std::vector<u64> ao(count + 1);
        HIPCHK(ctx, hipMemcpy(ao.data(), b->aoff + first, (count + 1) * 8, hipMemcpyDeviceToHost));
        const u64 nb = ao[count] - ao[0];
        if (nb > bytes_cap) return fail_arg(ctx, "bsk_batch_fetch_ascii: bytes_cap too small");
        if (nb) HIPCHK(ctx, hipMemcpy(bytes, b->ascii + ao[0], nb, hipMemcpyDeviceToHost));
        for (u64 i = 0; i <= count; ++i) offsets[i] = ao[i] - ao[0];
        return BSK_OK;
    }
    std::vector<u64> d(count), fwv(count);
    if (b->desc) {
        HIPCHK(ctx, hipMemcpy(d.data(), b->desc + first, count * 8, hipMemcpyDeviceToHost));
        for (u64 i = 0; i < count; ++i) {
            fwv[i] = d[i] >> 24;
            d[i] &= 0xffffffULL;
        }
    } else {
        HIPCHK(ctx, hipMemcpy(fwv.data(), b->fw + first, count * 8, hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(d.data(), b->llen + first, count * 8, hipMemcpyDeviceToHost));
    }
    const u64 w0 = fwv[0];
    const u64 w1 = fwv[count - 1] + (d[count - 1] + 15) / 16;
    std::vector<u32> w(w1 - w0 + 1);
    if (w1 > w0) HIPCHK(ctx, hipMemcpy(w.data(), b->words + w0, (w1 - w0) * 4, hipMemcpyDeviceToHost));
    u64 o = 0;
    for (u64 i = 0; i < count; ++i) {
        const u64 L = d[i], base = fwv[i] - w0;
        if (o + L > bytes_cap) return fail_arg(ctx, "bsk_batch_fetch_ascii: bytes_cap too small");
        for (u64 p = 0; p < L; ++p) bytes[o + p] = "ACGT"[(w[base + (p >> 4)] >> ((p & 15) * 2)) & 3];
        o += L;
        offsets[i + 1] = o;
    }
    return BSK_OK;
}

// ------------------------------------------------------------------------------------
// results
// ------------------------------------------------------------------------------------
// ---- spare result arrays (bsk_ctx::spare) ----------------------------------------------------------------------------------------------
static std::mutex g_spare_mu;            // (contexts are single-threaded; the out-of-memory flush walks every live context)
static std::vector<bsk_ctx *> g_ctxs;    // live contexts
void spare_register(bsk_ctx *ctx, bool add) {
    std::lock_guard<std::mutex> lk(g_spare_mu);
    if (add) g_ctxs.push_back(ctx);
    else g_ctxs.erase(std::remove(g_ctxs.begin(), g_ctxs.end(), ctx), g_ctxs.end());
}
static void spare_flush_locked(bsk_ctx *ctx) {
    for (auto &s : ctx->spare) {
        if (s.p) (void)hipFree(s.p);
        s = bsk_ctx::Spare();
    }
    ctx->spare_bytes = 0;
}
void spare_flush(bsk_ctx *ctx) {
    std::lock_guard<std::mutex> lk(g_spare_mu);
    spare_flush_locked(ctx);
}
// an allocation of the library failed: everything every context keeps in reserve goes back to the device (host_internal.hpp: then once more)
bool spare_flush_all() {
    std::lock_guard<std::mutex> lk(g_spare_mu);
    bool any = false;
    int dev = 0;
    (void)hipGetDevice(&dev);
    for (bsk_ctx *c : g_ctxs) {
        if (!c->spare_bytes) continue;
        any = true;
        (void)hipSetDevice(c->device);
        spare_flush_locked(c);
    }
    (void)hipSetDevice(dev);
    return any;
}
// a released result's array: kept for the next result (true) or freed
void spare_give(bsk_ctx *ctx, void *p) {
    if (!p) return;
    size_t bytes = 0;
    if (!ctx || ctx->opt.no_spare || hipMemPtrGetInfo(p, &bytes) != hipSuccess || bytes < (1u << 20)) {
        (void)hipGetLastError();
        (void)hipFree(p);
        return;
    }
    std::lock_guard<std::mutex> lk(g_spare_mu);
    if (!ctx->spare_limit) {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) tot = (size_t)64 << 30;
        ctx->spare_limit = tot / 5 * 2;
    }
    int slot = -1, smallest = 0;
    for (int i = 0; i < bsk_ctx::SPARE_SLOTS; ++i) {
        if (!ctx->spare[i].p && slot < 0) slot = i;
        if (ctx->spare[i].bytes < ctx->spare[smallest].bytes) smallest = i;
    }
    if (slot < 0 && ctx->spare[smallest].bytes < bytes) {  // full: the smallest one makes room for a larger one
        (void)hipFree(ctx->spare[smallest].p);
        ctx->spare_bytes -= ctx->spare[smallest].bytes;
        ctx->spare[smallest] = bsk_ctx::Spare();
        slot = smallest;
    }
    if (slot < 0 || ctx->spare_bytes + bytes > ctx->spare_limit) {
        (void)hipFree(p);
        return;
    }
    ctx->spare[slot].p = p;
    ctx->spare[slot].bytes = bytes;
    ctx->spare_bytes += bytes;
}
// an array of at least `bytes` for a result: the best fit among the spare ones (none wastes more than half of itself), else hipMalloc
hipError_t spare_take(bsk_ctx *ctx, void **out, size_t bytes) {
    *out = nullptr;
    if (ctx && !ctx->opt.no_spare) {
        std::lock_guard<std::mutex> lk(g_spare_mu);
        int best = -1;
        for (int i = 0; i < bsk_ctx::SPARE_SLOTS; ++i) {
            const auto &s = ctx->spare[i];
            if (s.p && s.bytes >= bytes && s.bytes <= 2 * bytes + ((size_t)64 << 20) && (best < 0 || s.bytes < ctx->spare[best].bytes)) best = i;
        }
        if (best >= 0) {
            *out = ctx->spare[best].p;
            ctx->spare_bytes -= ctx->spare[best].bytes;
            ctx->spare[best] = bsk_ctx::Spare();
            return hipSuccess;
        }
    }
    return hipMalloc(out, bytes);
}

extern "C" void bsk_result_release(bsk_result *r) {
    if (!r) return;
    if (r->ctx) (void)hipSetDevice(r->ctx->device);
    if (r->classes) class_set_free(r->classes);
    if (r->ctx && r->ctx->cls_owner == r) r->ctx->cls_owner = nullptr;
    spare_give(r->ctx, r->refs);
    spare_give(r->ctx, r->wfirst);
    spare_give(r->ctx, r->wcount);
    spare_give(r->ctx, r->status);
    if (!r->arrays_borrowed) {
        spare_give(r->ctx, r->hash);
        spare_give(r->ctx, r->pos);
    }
    delete r;
}

bool kind_has_pos(int kind) { return kind == BSK_MINIMIZER || kind == BSK_SYNCMER || kind == BSK_PROT_MINIMIZER; }

// tail: tuples reserved BEHIND the logical capacity `cap` (class plans: the slabs of the adopted parts live there; the kernels of the
// result itself never see them)
int result_prepare(bsk_ctx *ctx, bsk_result **res, u64 n, int kind, u64 cap, u64 tail) {
    bsk_result *r = *res;
    const int hp = kind_has_pos(kind) ? 1 : 0;
    if (r && (r->ctx != ctx || r->n_cap < n || !r->refs)) {  // too small (or a wide result): start over
        bsk_result_release(r);
        r = nullptr;
        *res = nullptr;
    }
    if (!r) {
        r = new (std::nothrow) bsk_result();
        if (!r) return BSK_ERR_NOMEM;
        r->ctx = ctx;
        r->n_cap = n;
        hipError_t e;
        if ((e = spare_take(ctx, (void **)&r->refs, (n ? n : 1) * 8)) != hipSuccess || (e = spare_take(ctx, (void **)&r->status, n ? n : 1)) != hipSuccess) {
            bsk_result_release(r);
            return fail_hip(ctx, e, "result alloc");
        }
        *res = r;
    }
    r->n = n;
    r->has_pos = hp;
    r->kind = kind;
    r->main_cap = 0;
    r->ovf_cap = 0;
    if (!hp && r->pos) {  // a reused buffer of a position kind: implicit positions mean pos == NULL
        (void)hipFree(r->pos);
        r->pos = nullptr;
    }
    if (r->alloc_cap < cap + tail || r->arrays_borrowed || (hp && !r->pos)) {
        if (!r->arrays_borrowed) {
            spare_give(ctx, r->hash);
            spare_give(ctx, r->pos);
        }
        r->arrays_borrowed = false;
        r->hash = nullptr;
        r->pos = nullptr;
        r->cap = 0;
        r->alloc_cap = 0;
        hipError_t e;
        if ((e = spare_take(ctx, (void **)&r->hash, (cap + tail + 2) * 8)) != hipSuccess) return fail_hip(ctx, e, "result hash alloc");
        if (hp && (e = spare_take(ctx, (void **)&r->pos, (cap + tail + 2) * 4)) != hipSuccess) return fail_hip(ctx, e, "result pos alloc");
        r->alloc_cap = cap + tail;
    }
    r->tail_cap = tail;
    r->cap = r->alloc_cap - tail;  // (a re-used, larger allocation: the logical capacity grows with it, the tail stays at the end)
    return BSK_OK;
}

extern "C" int bsk_result_info(const bsk_result *r, uint64_t *n_reads, uint64_t *n_tuples, int *has_pos) {
    if (!r) return BSK_ERR_ARG;
    if (n_reads) *n_reads = r->n;
    if (n_tuples) *n_tuples = r->n_tuples;
    if (has_pos) *has_pos = r->has_pos;
    return BSK_OK;
}

extern "C" int bsk_result_plan(const bsk_result *r, const char **kernel, int *grid, int *waves_per_cu) {
    if (!r) return BSK_ERR_ARG;
    if (kernel) *kernel = r->plan;
    if (grid) *grid = r->plan_grid;
    if (waves_per_cu) *waves_per_cu = r->plan_per_cu;
    return BSK_OK;
}

extern "C" int bsk_result_device(const bsk_result *r, const uint64_t **refs, const uint8_t **status, const uint64_t **hash,
                                 const uint32_t **pos) {
    if (!r) return BSK_ERR_ARG;
    if (refs) *refs = (const uint64_t *)r->refs;  // NULL for wide results: bsk_result_device_wide
    if (status) *status = r->status;
    if (hash) *hash = (const uint64_t *)r->hash;
    if (pos) *pos = r->pos;
    return BSK_OK;
}

extern "C" int bsk_result_device_wide(const bsk_result *r, const uint64_t **first, const uint64_t **count) {
    if (!r) return BSK_ERR_ARG;
    if (first) *first = (const uint64_t *)r->wfirst;
    if (count) *count = (const uint64_t *)r->wcount;
    return BSK_OK;
}

// the status bytes alone (a consumer of sketch SETS still needs every read's SHORT / ILLEGAL / tie flags)
extern "C" int bsk_result_fetch_status(bsk_ctx *ctx, const bsk_result *r, uint64_t first, uint64_t count, uint8_t *status) {
    if (!ctx || !r || !status) return fail_arg(ctx, "bsk_result_fetch_status: null argument");
    if (first + count > r->n) return fail_arg(ctx, "bsk_result_fetch_status: range outside result");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (count) {
        HIPCHK(ctx, hipMemcpyAsync(status, r->status + first, count, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return BSK_OK;
}

extern "C" int bsk_result_fetch(bsk_ctx *ctx, const bsk_result *r, uint64_t first, uint64_t count, uint64_t *offsets,
                                uint8_t *status, uint64_t *hash, uint32_t *pos, uint64_t tuple_cap) {
    if (!ctx || !r || !offsets) return fail_arg(ctx, "bsk_result_fetch: null argument");
    if (first + count > r->n) return fail_arg(ctx, "bsk_result_fetch: range outside result");
    if (pos && !r->pos) return fail_arg(ctx, "bsk_result_fetch: this kind has implicit positions");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // everything goes over the context's stream through grow-only buffers: no hipMalloc / hipFree (a device-wide
    // synchronisation) and no pageable staging per call -- a streaming caller fetches chunk after chunk
    if (ctx->h_refs_cap < count + 1) {
        if (ctx->h_refs) (void)hipHostFree(ctx->h_refs);
        ctx->h_refs = nullptr;
        ctx->h_refs_cap = 0;
        const size_t want = (count + 1) + (count + 1) / 4 + 64;
        HIPCHK(ctx, hipHostMalloc(&ctx->h_refs, want * 8));
        ctx->h_refs_cap = want;
    }
    u64 *refs = ctx->h_refs;
    if (count) {
        HIPCHK(ctx, hipMemcpyAsync(refs, (r->refs ? r->refs : r->wcount) + first, count * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (status) HIPCHK(ctx, hipMemcpyAsync(status, r->status + first, count, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    offsets[0] = 0;
    for (u64 i = 0; i < count; ++i) offsets[i + 1] = offsets[i] + (r->refs ? (refs[i] & 0xffffffULL) : refs[i]);
    const u64 T = offsets[count];
    if (!hash && !pos) return BSK_OK;
    if (T > tuple_cap) return fail_arg(ctx, "bsk_result_fetch: tuple_cap too small");
    if (T == 0) return BSK_OK;
    // pack on the device (the tuple arrays are slab-organised), then one D2H copy per array
    u64 *d_off = nullptr, *d_h = nullptr;
    u32 *d_p = nullptr;
    auto fpool = [&](int slot, size_t bytes, void **out) -> hipError_t {
        if (ctx->tmp_cap[slot] < bytes) {
            (void)hipFree(ctx->tmp[slot]);
            ctx->tmp[slot] = nullptr;
            ctx->tmp_cap[slot] = 0;
            const size_t want = bytes + bytes / 4 + 256;
            const hipError_t e2 = hipMalloc(&ctx->tmp[slot], want);
            if (e2 != hipSuccess) return e2;
            ctx->tmp_cap[slot] = want;
        }
        *out = ctx->tmp[slot];
        return hipSuccess;
    };
    hipError_t e = fpool(12, count * 8, (void **)&d_off);
    if (e == hipSuccess && hash) e = fpool(13, T * 8, (void **)&d_h);
    if (e == hipSuccess && pos) e = fpool(14, T * 4, (void **)&d_p);
    for (u64 i = 0; i < count; ++i) refs[i] = offsets[i];  // the pinned buffer carries the offsets back up
    if (e == hipSuccess) e = hipMemcpyAsync(d_off, refs, count * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_gather, dim3(grid_for(ctx, count * 64, 256)), dim3(256), 0, ctx->stream, r->hash, r->pos,
                           r->refs ? r->refs + first : nullptr, r->refs ? nullptr : r->wfirst + first,
                           r->refs ? nullptr : r->wcount + first, d_off, count, d_h, d_p);
        e = hipGetLastError();
    }
    if (e == hipSuccess && hash) e = hipMemcpyAsync(hash, d_h, T * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && pos) e = hipMemcpyAsync(pos, d_p, T * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail_hip(ctx, e, "bsk_result_fetch");
    return BSK_OK;
}

extern "C" int bsk_result_digest(bsk_ctx *ctx, const bsk_result *r, uint64_t *checksum, uint64_t *n_tuples,
                                 uint64_t status_counts[4]) {
    if (!ctx || !r) return fail_arg(ctx, "bsk_result_digest: null argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_total, 0, 8 * sizeof(u64), ctx->stream));
    if (r->n) {
        hipLaunchKernelGGL(k_digest, dim3(grid_for(ctx, r->n * 8, 256)), dim3(256), 0, ctx->stream, r->hash, r->pos, r->refs, r->wfirst,
                           r->wcount, r->n, ctx->d_total);
        hipLaunchKernelGGL(k_digest_status, dim3(grid_for(ctx, r->n, 256)), dim3(256), 0, ctx->stream, r->status, r->n,
                           ctx->d_total + 2);
    }
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_pinned, ctx->d_total, 8 * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (checksum) *checksum = ctx->h_pinned[0];
    if (n_tuples) *n_tuples = ctx->h_pinned[1];
    if (status_counts)
        for (int i = 0; i < 4; ++i) status_counts[i] = ctx->h_pinned[2 + i];
    return BSK_OK;
}


// ------------------------------------------------------------------------------------
// DNA/RNA -> protein (the Translate call of NewProteinIterator / NewProteinMinimizerSketch,
// iterator-protein.go:62-67, sketch-protein.go:83-88)
// ------------------------------------------------------------------------------------
// NCBI genetic codes: the standard code plus each table's reassigned codons (codon order T,C,A,G; the ids are the
// ones registered at seq/codon_tables.go:431-621).
static const char kStdCode[65] = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
static const struct { int id; const char *diff; } kGeneticCodes[] = {
    {1, ""}, {2, "TGAW ATAM AGA* AGG*"}, {3, "TGAW CTTT CTCT CTAT CTGT ATAM"}, {4, "TGAW"}, {5, "TGAW ATAM AGAS AGGS"},
    {6, "TAAQ TAGQ"}, {9, "TGAW AAAN AGAS AGGS"}, {10, "TGAC"}, {11, ""}, {12, "CTGS"}, {13, "TGAW ATAM AGAG AGGG"},
    {14, "TAAY TGAW AAAN AGAS AGGS"}, {16, "TAGL"}, {21, "TGAW ATAM AAAN AGAS AGGS"}, {22, "TCA* TAGL"}, {23, "TTA*"},
    {24, "TGAW AGAS AGGK"}, {25, "TGAG"}, {26, "CTGA"}, {27, "TAAQ TAGQ TGAW"}, {28, "TAAQ TAGQ TGAW"}, {29, "TAAY TAGY"},
    {30, "TAAE TAGE"}, {31, "TAAE TAGE TGAW"},
};
#define BSK_LUT_BYTES (4096 + 256 + 64)

// IUPAC letter -> 4-bit base set (A1 C2 G4 T/U8), gap letters 0, everything else 16 (seq/ambiguous_bases.go:28-67)
static unsigned iupac_set(unsigned b) {
    static const char letters[] = "ACMGRSVTWYHKDBN";  // letter of set 1..15
    if (b == ' ' || b == '*' || b == '-') return 0;
    if (b == 'U' || b == 'u') return 8;
    for (unsigned c = 1; c < 16; ++c)
        if (b == (unsigned)letters[c - 1] || b == (unsigned)(letters[c - 1] | 0x20)) return c;
    return 16;
}

// Tables for kernels_translate.hpp.  A codon made of base SETS has an amino acid iff every plain codon it stands for has
// that amino acid -- the fixed point of the three passes of codonTableFromText (seq/codon_tables.go:350-427); entries
// left empty there read as 'X' (Get, :172-174).  Pure host code.
extern "C" int bsk_codon_lut(int table, uint8_t *lut, uint64_t lut_bytes) {
    if (!lut || lut_bytes < BSK_LUT_BYTES) return BSK_ERR_ARG;
    const char *diff = nullptr;
    for (const auto &g : kGeneticCodes)
        if (g.id == table) diff = g.diff;
    if (!diff) return BSK_ERR_ARG;
    char aa[64];
    memcpy(aa, kStdCode, 64);
    auto tcag = [](char c) { return c == 'T' ? 0 : c == 'C' ? 1 : c == 'A' ? 2 : 3; };
    for (const char *d = diff; *d; d += d[4] ? 5 : 4) aa[tcag(d[0]) * 16 + tcag(d[1]) * 4 + tcag(d[2])] = d[3];
    static const int bit_to_tcag[9] = {-1, 2, 1, -1, 3, -1, -1, -1, 0};  // set bit A1 C2 G4 T8 -> index in T,C,A,G order
    for (unsigned i = 0; i < 16; ++i)
        for (unsigned j = 0; j < 16; ++j)
            for (unsigned k = 0; k < 16; ++k) {
                int common = -1;  // -1 nothing yet, 0 disagreement
                if (i && j && k)
                    for (unsigned a = 1; a <= 8 && common != 0; a <<= 1)
                        for (unsigned b = 1; b <= 8 && common != 0; b <<= 1)
                            for (unsigned c = 1; c <= 8 && common != 0; c <<= 1) {
                                if (!(i & a) || !(j & b) || !(k & c)) continue;
                                const int v = aa[bit_to_tcag[a] * 16 + bit_to_tcag[b] * 4 + bit_to_tcag[c]];
                                common = common < 0 ? v : (common == v ? v : 0);
                            }
                lut[(i << 8) | (j << 4) | k] = common > 0 ? (uint8_t)common : (uint8_t)'X';
            }
    for (unsigned b = 0; b < 256; ++b) lut[4096 + b] = (uint8_t)iupac_set(b);
    static const int acgt_to_tcag[4] = {2, 1, 3, 0};  // 2-bit code A0 C1 G2 T3
    for (unsigned c = 0; c < 64; ++c)
        lut[4096 + 256 + c] = (uint8_t)aa[acgt_to_tcag[c >> 4] * 16 + acgt_to_tcag[(c >> 2) & 3] * 4 + acgt_to_tcag[c & 3]];
    return BSK_OK;
}

// Translate every sequence of a DNA batch into a new protein batch.  need != 0: sequences shorter than `need` bases are
// flagged (rflags) so that the protein kernels report them as ErrShortSeq -- the reference checks the INPUT length.
// the codon tables of `table` on the device (kernels_translate.hpp layout), cached per context
static int ensure_lut(bsk_ctx *ctx, int table) {
    if (ctx->lut_table == table && ctx->d_lut) return BSK_OK;
    uint8_t lut[BSK_LUT_BYTES];
    if (bsk_codon_lut(table, lut, sizeof lut) != BSK_OK) {
        ctx->err = "invalid codon table";  // seq/seq.go:691
        return BSK_ERR_ARG;
    }
    if (!ctx->d_lut) HIPCHK(ctx, hipMalloc(&ctx->d_lut, BSK_LUT_BYTES));
    HIPCHK(ctx, hipMemcpy(ctx->d_lut, lut, BSK_LUT_BYTES, hipMemcpyHostToDevice));
    ctx->lut_table = table;
    return BSK_OK;
}

static int translate_batch(bsk_ctx *ctx, const bsk_batch *b, int table, int frame, u64 need, bsk_batch **out) {
    *out = nullptr;
    if (frame < -3 || frame > 3 || frame == 0) {
        ctx->err = "invalid frame (available: 1, 2, 3, -1, -2, -3)";  // seq/seq.go:694
        return BSK_ERR_ARG;
    }
    int rc0 = ensure_lut(ctx, table);
    if (rc0 != BSK_OK) return rc0;
    bsk_batch *t = new (std::nothrow) bsk_batch();
    if (!t) return BSK_ERR_NOMEM;
    t->ctx = ctx;
    t->alphabet = BSK_ALPHA_PROTEIN;
    t->n = b->n;
    t->maxlen = (u32)translated_len(b->maxlen, frame > 0 ? 1 : -1);
    t->uniform_len = b->uniform_len ? (u32)translated_len(b->uniform_len, frame) : 0;
    const u64 cap_bytes = b->n_bases / 3 + 1;
    hipError_t e;
    if ((e = hipMalloc(&t->ascii, cap_bytes + BSK_ASCII_PAD)) != hipSuccess || (e = hipMalloc(&t->aoff, (b->n + 1) * 8)) != hipSuccess ||
        (e = hipMalloc(&t->rflags, b->n ? b->n : 1)) != hipSuccess ||
        (e = hipMemsetAsync(t->aoff, 0, 8, ctx->stream)) != hipSuccess) {
        bsk_batch_destroy(t);
        return fail_hip(ctx, e, "translate alloc");
    }
    t->device_bytes = cap_bytes + BSK_ASCII_PAD + (b->n + 1) * 8 + b->n;
    const u32 nunits = (u32)((b->n + 63) / 64);
    if (nunits) {
        int rc = ensure_scratch(ctx, lb_words_with_heads(nunits), 0);  // (k_translate takes its units from eight ticket heads behind the granules)
        if (rc != BSK_OK) {
            bsk_batch_destroy(t);
            return rc;
        }
        TArgs a;
        memset(&a, 0, sizeof a);
        a.words = b->words;
        a.desc = b->desc;
        a.fw = b->fw;
        a.llen = b->llen;
        a.ascii = b->ascii;
        a.aoff = b->aoff;
        a.n = b->n;
        a.nunits = nunits;
        a.frame = frame;
        a.need = need;
        a.lut = ctx->d_lut;
        a.out = t->ascii;
        a.out_off = t->aoff;
        a.short_flag = t->rflags;
        a.ticket = ctx->d_ticket;
        a.lookback = ctx->d_lookback;
        a.total = ctx->d_total;
        const bool use_ascii = b->n_nonacgt > 0;
        int per_cu = use_ascii ? blocks_per_cu(k_translate<1>) : blocks_per_cu(k_translate<0>);
        const int grid = (int)std::max<u64>(1, std::min<u64>((u64)ctx->cus * per_cu, nunits));
        if ((e = hipMemsetAsync(ctx->d_ticket, 0, 2 * sizeof(u32), ctx->stream)) == hipSuccess &&
            (e = hipMemsetAsync(ctx->d_total, 0, 2 * sizeof(u64), ctx->stream)) == hipSuccess &&
            (e = hipMemsetAsync(ctx->d_lookback, 0, lb_words_with_heads(nunits) * sizeof(u64), ctx->stream)) == hipSuccess) {
            if (use_ascii) hipLaunchKernelGGL(k_translate<1>, dim3(grid), dim3(64), 0, ctx->stream, a);
            else hipLaunchKernelGGL(k_translate<0>, dim3(grid), dim3(64), 0, ctx->stream, a);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(ctx->h_pinned, ctx->d_total, sizeof(u64), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            bsk_batch_destroy(t);
            return fail_hip(ctx, e, "translate");
        }
        t->n_bases = ctx->h_pinned[0];
    }
    *out = t;
    return BSK_OK;
}

extern "C" int bsk_batch_translate(bsk_ctx *ctx, const bsk_batch *dna, int codon_table, int frame, bsk_batch **out) {
    if (!ctx || !dna || !out) return fail_arg(ctx, "bsk_batch_translate: null argument");
    if (dna->ctx != ctx) return fail_arg(ctx, "bsk_batch_translate: batch belongs to another context");
    if (dna->alphabet != BSK_ALPHA_DNA) return fail_arg(ctx, "bsk_batch_translate: only DNA/RNA batches can be translated");  // seq.go:686
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = translate_batch(ctx, dna, codon_table, frame, 0, out);
    if (rc == BSK_OK) {  // a stand-alone protein batch: length checks then apply to the protein, as for any Protein Seq
        (void)hipFree((*out)->rflags);
        (*out)->rflags = nullptr;
    }
    return rc;
}


extern "C" int bsk_batch_prepare(bsk_ctx *ctx, const bsk_batch *batch, const bsk_params *p, float *ms) {
    if (ms) *ms = 0.0f;
    if (!ctx || !batch || !p) return fail_arg(ctx, "bsk_batch_prepare: null argument");
    if (batch->ctx != ctx) return fail_arg(ctx, "bsk_batch_prepare: the batch belongs to another context");
    if (batch->n == 0 || p->circular || !batch->desc) return BSK_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    {  // a batch that takes a class plan: the pass that cuts it by length (its lists and the bulk's view live in the context's pool, so a
       // class-plan RESULT sized earlier on this context can no longer be re-run by bsk_sketch_timed -- bsk_sketch sizes it again)
        std::vector<ClassCut> cuts;
        int bulk = 0;
        if (class_decide(ctx, batch, p, 0, cuts, bulk)) {
            ClassSet *tmp = new (std::nothrow) ClassSet();
            if (!tmp) return BSK_ERR_NOMEM;
            ctx->cls_owner = nullptr;
            const int crc = class_build(ctx, batch, p, cuts, bulk, tmp);
            if (ms) *ms = tmp->build_ms;
            class_set_free(tmp);
            return crc;
        }
    }
    Plan pl;
    int rc = make_plan(ctx, batch, p, pl);
    if (rc != BSK_OK || !pl.bin_gran) return rc == BSK_OK ? BSK_OK : BSK_OK;  // (parameters bsk_sketch would refuse are its to report)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIPCHK(ctx, hipEventCreate(&e0));
    {
        const hipError_t ec = hipEventCreate(&e1);
        if (ec != hipSuccess) {
            (void)hipEventDestroy(e0);
            return fail_hip(ctx, ec, "bsk_batch_prepare: hipEventCreate");
        }
    }
    if (batch->bin_early) {  // the view came with the batch: no pass per plan
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        return BSK_OK;
    }
    batch->bin_gran = 0;  // build (again): the call is also the way to time the pass
    hipError_t e = hipEventRecord(e0, ctx->stream);
    rc = ensure_binned(ctx, batch, (u32)((p->kind == BSK_SYNCMER ? p->s : p->k) - 1), pl.bin_gran, 0, 0, 0);
    if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float t = 0.0f;
    if (e == hipSuccess) e = hipEventElapsedTime(&t, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != BSK_OK) return rc;
    if (e != hipSuccess) return fail_hip(ctx, e, "bsk_batch_prepare");
    if (ms) *ms = t;
    return BSK_OK;
}

static int sketch_impl(bsk_ctx *ctx, const bsk_batch *batch, const bsk_params *p, bsk_result **result, int warmup, int iters,
                       float *kernel_ms) {
    if (!ctx || !batch || !p || !result) return fail_arg(ctx, "bsk_sketch: null argument");
    if (batch->ctx != ctx) return fail_arg(ctx, "bsk_sketch: batch belongs to another context");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = validate(p, batch->alphabet);
    if (rc != BSK_OK) {
        ctx->err = bsk_err_name(rc);
        return rc;
    }
    // a syncmer sketch with s == k yields every k-mer with its index (sketch.go:328-331) -- the minimizer sketch with w = 1 (sketch.go:
    // 218-222), same length rule (sketch.go:179-182): it runs as that (k_minimizer_dense<1>; the tiled path always did, and until round 5
    // shorter sequences took the general syncmer kernel at 50-70 Gbases/s).  Not circular sequences: their length rules differ.
    bsk_params as_min;
    if (p->kind == BSK_SYNCMER && p->s == p->k && !p->circular && batch->alphabet == BSK_ALPHA_DNA && !ctx->opt.force_generic) {
        as_min = *p;
        as_min.kind = BSK_MINIMIZER;
        as_min.w = 1;
        p = &as_min;
    }
    const bsk_batch *b = batch;
    bsk_batch *tmp = nullptr;
    int circ_ext = 0;
    const bool prot_kind = p->kind == BSK_PROT_HASH || p->kind == BSK_PROT_MINIMIZER;
    bool fused = false;
    if (batch->alphabet == BSK_ALPHA_DNA && prot_kind) {
        // iterator-protein.go:50,62-67 / sketch-protein.go:66-75,83-88: length checks on the nucleotides, then Translate
        if (p->frame < -3 || p->frame > 3 || p->frame == 0) {
            ctx->err = "invalid frame (available: 1, 2, 3, -1, -2, -3)";  // seq/seq.go:694
            return BSK_ERR_ARG;
        }
        // pure-ACGT 2-bit batches and a compiled (w, k): the protein minimizer kernel translates on the fly (no translated copy)
        fused = p->kind == BSK_PROT_MINIMIZER && batch->desc && batch->n_nonacgt == 0 && fast_prot_supported(p->w, p->k) &&
                translated_len(batch->maxlen, 1) < 65536u && !ctx->opt.force_generic && !ctx->opt.no_fused_translate && !ctx->no_prot_fast &&
                slab_budget_ok(batch, (u64)translated_len(batch->maxlen, 1));
        // the hash stream likewise (k = 9..16; translations longer than the tile threshold take the tiled two-step path)
        if (p->kind == BSK_PROT_HASH)
            fused = batch->desc && batch->n_nonacgt == 0 && fast_prot_hash_supported(p->k) && translated_len(batch->maxlen, 1) <= 4096u &&
                    !ctx->opt.force_generic && !ctx->opt.no_fused_translate;
        if (fused) {
            rc = ensure_lut(ctx, p->codon_table);
            if (rc != BSK_OK) return rc;
        } else {
            const u64 need = (u64)p->k * 3 + (p->kind == BSK_PROT_MINIMIZER ? (u64)p->w - 1 : 0);
            rc = translate_batch(ctx, batch, p->codon_table, p->frame, need, &tmp);
            if (rc != BSK_OK) return rc;
            b = tmp;
        }
    } else if (p->circular && p->k > 1 && batch->alphabet == BSK_ALPHA_DNA) {
        rc = make_circular(ctx, batch, p->k, &tmp);
        if (rc != BSK_OK) return rc;
        b = tmp;
        circ_ext = p->k - 1;
    }
    // the two-strand k-mer mode (iterator.go:713-723) yields 2(L-k+1) values per read: without tiles (dev switch) a read's count must
    // fit the 24-bit field of its reference word
    if (p->kind == BSK_KMER && !p->canonical && ctx->opt.no_tiles && (u64)b->maxlen >= (1ull << 23) + (u64)p->k - 1) {
        if (tmp) bsk_batch_destroy(tmp);
        ctx->err = "two-strand k-mer codes: sequences of 2^23 k-mers or more need the tiled path";
        return BSK_ERR_UNSUPPORTED;
    }
    // tile descriptors keep the tile length (~ positions + 2w + k) in 24 bits, and the generic kernels size their window ring by w
    if ((u64)p->k >= (1ull << 22) || ((p->kind == BSK_MINIMIZER || p->kind == BSK_PROT_MINIMIZER) && (u64)p->w >= (1ull << 22)) ||
        (p->kind == BSK_SYNCMER && (u64)(p->k - p->s) >= (1ull << 21))) {
        // such a window cannot fit a sequence below 2^24 bases with room to select anything; longer sequences would need wider descriptors
        if (b->maxlen >= (1u << 22)) {
            if (tmp) bsk_batch_destroy(tmp);
            ctx->err = "k / w of 2^22 or more on sequences of 2^22 bases or more is not supported";
            return BSK_ERR_UNSUPPORTED;
        }
    }
    // long sequences run as tiles; protein: only when really long (the protein kernels take any length per lane, slowly)
    const bool is_dna = b->alphabet == BSK_ALPHA_DNA;
    const u32 tile_min = tile_min_for(ctx, b, p);
    if (!tmp) {  // a batch of several length classes: one plan per class (run_classed), outliers cost their own bases
        bool applied = false;
        rc = run_classed(ctx, b, p, circ_ext, result, warmup, iters, kernel_ms, &applied);
        if (rc != BSK_OK || applied) return rc;
    }
    const bool outlier = !is_dna && p->kind == BSK_PROT_MINIMIZER && b->maxlen > 512 && !slab_budget_ok(b, (u64)b->maxlen);  // tiles are uniform: small slabs
    const bool tiled = kind_tiles(p) && (is_dna != prot_kind) && !ctx->opt.no_tiles && ((is_dna && !b->desc) || b->maxlen > tile_min || outlier);
    rc = tiled ? sketch_tiled(ctx, b, p, circ_ext, result, warmup, iters, kernel_ms)
               : run_planned_resizing(ctx, b, p, circ_ext, result, warmup, iters, kernel_ms);
    if (rc == BSK_REPLAN_UNFUSED && fused) {  // unusual density (a sequence outgrew its slab) or slabs that do not fit: translate, then sketch
        ctx->no_prot_fast = false;
        const u64 need = (u64)p->k * 3 + (u64)p->w - 1;
        rc = translate_batch(ctx, batch, p->codon_table, p->frame, need, &tmp);
        if (rc != BSK_OK) return rc;
        const bool tiled2 = kind_tiles(p) && !ctx->opt.no_tiles && tmp->maxlen > tile_min;
        rc = tiled2 ? sketch_tiled(ctx, tmp, p, 0, result, warmup, iters, kernel_ms) : run_planned_resizing(ctx, tmp, p, 0, result, warmup, iters, kernel_ms);
    }
    if (tmp) bsk_batch_destroy(tmp);
    return rc;
}

static int public_rc(bsk_ctx *ctx, int rc) {  // the internal re-plan codes never cross the boundary
    if (rc > -1000) return rc;
    if (ctx) ctx->err = "internal: a re-plan request reached the boundary (code " + std::to_string(rc) + ")";
    return BSK_ERR_DEVICE;
}
extern "C" int bsk_sketch(bsk_ctx *ctx, const bsk_batch *batch, const bsk_params *p, bsk_result **result) {
    return public_rc(ctx, sketch_impl(ctx, batch, p, result, 0, 0, nullptr));
}

extern "C" int bsk_sketch_timed(bsk_ctx *ctx, const bsk_batch *batch, const bsk_params *p, bsk_result **result, int warmup,
                                int iters, float *kernel_ms) {
    if (warmup < 0 || iters < 0) return fail_arg(ctx, "bsk_sketch_timed: negative counts");
    return public_rc(ctx, sketch_impl(ctx, batch, p, result, warmup, iters, kernel_ms));
}

// circular=true: build a temporary batch whose reads carry their first k-1 bases appended
// (iterator.go:642-646, sketch.go:106-110,163-167).
int make_circular(bsk_ctx *ctx, const bsk_batch *b, int k, bsk_batch **out) {
    *out = nullptr;
    const u64 n = b->n;
    // lengths of the sequences: from the packed descriptors, or (a batch with a sequence of 2^24 bases or more) from llen
    std::vector<u64> desc(n ? n : 1), nd(n ? n : 1), nl;
    if (n) HIPCHK(ctx, hipMemcpy(desc.data(), b->desc ? b->desc : b->llen, n * 8, hipMemcpyDeviceToHost));
    bsk_batch *t = new (std::nothrow) bsk_batch();
    if (!t) return BSK_ERR_NOMEM;
    t->ctx = ctx;
    t->alphabet = b->alphabet;
    t->pairs = b->pairs;
    t->n = n;
    u64 w = 0, nb = 0;
    u32 maxlen = 0;
    std::vector<u64> nao(n + 1);
    bool wide = false;  // the extended batch needs first-word / length arrays (it will run as tiles)
    for (u64 r = 0; r < n; ++r) {
        const u64 L = b->desc ? (desc[r] & 0xffffffULL) : desc[r];
        const u64 ext = std::min<u64>(L, (u64)(k - 1));  // reads shorter than k-1 are ErrShortSeq anyway
        if (L + ext >= (1ULL << 24)) wide = true;
    }
    if (wide) nl.resize(n);
    for (u64 r = 0; r < n; ++r) {
        const u64 L = b->desc ? (desc[r] & 0xffffffULL) : desc[r];
        const u64 ext = std::min<u64>(L, (u64)(k - 1));
        const u64 L2 = L + ext;
        if (L2 >= (1ULL << 31)) {
            delete t;
            ctx->err = "circular sequence too long (2^31 bases with its k-1 appended bases)";
            return BSK_ERR_UNSUPPORTED;
        }
        if (wide) {
            nd[r] = w;
            nl[r] = L2;
        } else {
            nd[r] = (w << 24) | L2;
        }
        nao[r] = nb;
        w += (L2 + 15) / 16;
        nb += L2;
        maxlen = std::max<u32>(maxlen, (u32)L2);
    }
    nao[n] = nb;
    t->n_bases = nb;
    t->n_words = w;
    t->maxlen = maxlen;
    t->n_nonacgt = b->n_nonacgt;
    t->uniform_len = (b->uniform_len && b->uniform_len >= (u32)(k - 1)) ? b->uniform_len + (u32)(k - 1) : 0;
    const u64 alloc_words = w + pad_words(maxlen);
    hipError_t e;
    if ((e = hipMalloc(&t->words, alloc_words * 4)) != hipSuccess || (e = hipMalloc(wide ? &t->fw : &t->desc, (n ? n : 1) * 8)) != hipSuccess ||
        (wide && (e = hipMalloc(&t->llen, (n ? n : 1) * 8)) != hipSuccess) ||
        (e = hipMalloc(&t->rflags, n ? n : 1)) != hipSuccess ||
        (e = hipMemsetAsync(t->words, 0, alloc_words * 4, ctx->stream)) != hipSuccess ||
        (n && (e = hipMemcpyAsync(wide ? t->fw : t->desc, nd.data(), n * 8, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) ||
        (n && wide && (e = hipMemcpyAsync(t->llen, nl.data(), n * 8, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) ||
        (n && (e = hipMemcpyAsync(t->rflags, b->rflags, n, hipMemcpyDeviceToDevice, ctx->stream)) != hipSuccess)) {
        bsk_batch_destroy(t);
        return fail_hip(ctx, e, "make_circular alloc");
    }
    if (n && w)
        hipLaunchKernelGGL(k_extend_packed, dim3(grid_for(ctx, w, 256)), dim3(256), 0, ctx->stream, b->words, b->desc, b->fw, b->llen, t->desc,
                           t->fw, t->llen, n, w, t->words);

    if (b->ascii && n) {  // batches with non-ACGT bytes are hashed from ASCII: extend that too
        if ((e = hipMalloc(&t->ascii, nb + BSK_ASCII_PAD)) != hipSuccess || (e = hipMalloc(&t->aoff, (n + 1) * 8)) != hipSuccess ||
            (e = hipMemcpyAsync(t->aoff, nao.data(), (n + 1) * 8, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) {
            bsk_batch_destroy(t);
            return fail_hip(ctx, e, "make_circular ascii alloc");
        }
        hipLaunchKernelGGL(k_extend_ascii, dim3(grid_for(ctx, n * 64, 256)), dim3(256), 0, ctx->stream, b->ascii, b->aoff, t->aoff,
                           n, t->ascii);
    }
    e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) {
        bsk_batch_destroy(t);
        return fail_hip(ctx, e, "make_circular");
    }
    if (t->ascii) {
        const int rc = build_subset(ctx, t);
        if (rc != BSK_OK) {
            bsk_batch_destroy(t);
            return rc;
        }
    }
    *out = t;
    return BSK_OK;
}

