// launch.hip -- one launch of a planned kernel (launch), and plan + size + launch + time (run_planned): the only translation unit that
// instantiates the general, stream and SimHash kernel templates (kernels_generic / _fast / _more / _simhash .hpp).
#include "host_internal.hpp"
#include "kernels_host.hpp"

// One launch of the planned kernel into res.  ev0/ev1 (optional) bracket the kernel itself (with the parts of a class plan).
int launch(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p, bsk_result *res, int circ_ext, const Plan &pl,
                  hipEvent_t ev0, hipEvent_t ev1) {
    if (pl.nunits == 0) return BSK_OK;
    ClassSet *const cs = (ctx->cls && b == ctx->cls->view) ? ctx->cls : nullptr;
    if (cs) {
        if (ev0) HIPCHK(ctx, hipEventRecord(ev0, ctx->stream));
        // the side stream starts where the main stream is now (the lists of the parts' reads, the previous launch's adoption of the parts'
        // reference words), then takes the one-launch parts
        HIPCHK(ctx, hipMemsetAsync(ctx->d_ticket + 16, 0, sizeof(u32), ctx->stream));  // the parts' overflow flags of THIS launch (k_fold_flags)
        HIPCHK(ctx, hipEventRecord(ctx->ev_adopted, ctx->stream));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->side->stream, ctx->ev_adopted, 0));
        // tiled parts first: sketch_tiled waits for its counts on the host, and queued behind the bulk's kernel its launches would only start
        // when the bulk's persistent waves retire (they hold every CU's LDS) -- measured: 0.835 against 0.851 of the uniform rate
        const int trc = launch_parts(ctx, cs, p, res, true);
        if (trc != BSK_OK) return trc;
        // (round 6: the tiled parts no longer wait on the host -- but their CHAIN of small kernels must be through before the bulk's persistent
        // waves take every CU, or its later links only run when those retire: 0.855 of the uniform rate against 0.908 with the host waits)
        bool any_tiled = false;
        for (auto &pt : cs->parts) any_tiled |= pt.tiled && pt.n;
        if (any_tiled) {
            HIPCHK(ctx, hipEventRecord(ctx->ev_tiled, ctx->side->stream));
            HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_tiled, 0));
        }
        const int prc = launch_parts(ctx, cs, p, res, false);
        if (prc != BSK_OK) return prc;
    }
    KArgs a;
    memset(&a, 0, sizeof a);
    a.words = b->words;
    a.desc = b->desc;
    a.ascii = b->ascii;
    a.aoff = b->aoff;
    a.adesc = b->adesc;
    a.rflags = b->rflags;
    a.n = b->n;
    a.nunits = pl.nunits;
    a.kind = p->kind;
    a.k = p->k;
    a.w = p->w;
    a.s = p->s;
    a.m = p->m;
    a.scale = p->scale;
    a.canonical = p->canonical > 0 ? 1 : 0;
    a.one_strand = p->canonical < 0 ? 1 : 0;  // sketch_tiled's internal value
    a.pairs = b->pairs;
    a.circ_ext = circ_ext;
    a.uniform_len = b->uniform_len;
    a.refs = res->refs;
    a.status = res->status;
    a.hash = res->hash;
    a.pos = res->pos;
    a.cap = res->main_cap ? res->main_cap : res->cap;
    a.ovf_base = pl.slab_total;
    a.slab_read = pl.slab_read;
    a.ovf_cap = res->ovf_cap;
    a.ticket = ctx->d_ticket;
    a.total = ctx->d_total;
    a.ring_w = pl.ring_w;
    a.len_mask = 0xffffffu;
    {  // units per ticket (KArgs::tk): the kernel's own while every wavefront of the grid gets a whole ticket, fewer below that
        const u32 own = (pl.which == K_MIN_PKD || pl.which == K_MIN_DENSE || pl.which == K_PROT_MIN_FAST) ? 4u : 8u;
        const u64 waves = (u64)std::max(pl.grid, 1);
        a.tk = 0;
        if ((pl.which == K_MIN_PK || pl.which == K_MIN_PKD || pl.which == K_MIN_RING || pl.which == K_SYN_PK || pl.which == K_MIN_FAST || pl.which == K_MIN_DENSE || pl.which == K_SYN_FAST ||
             pl.which == K_PROT_MIN_FAST) && (u64)pl.nunits < waves * own)
            a.tk = (u32)std::max<u64>(1, ((u64)pl.nunits + waves - 1) / waves);  // (rounded up: one ticket per wavefront -- rounded down, 10^6 reads were 2 232 tickets of seven units on 2 048 wavefronts)
    }
    if (cs && cs->masked) {  // class plan without a view: the kernel masks the other classes' reads itself (desc_len)
        a.cls_lo = cs->blo;
        a.cls_hi = cs->bhi;
        a.cls_pretend = cs->pretend;
    }
    if (pl.bin_gran) {  // ragged short reads on a lock-step kernel: units of reads that end together (k_bin_desc)
        const int brc = ensure_binned(ctx, b, (u32)((p->kind == BSK_SYNCMER ? p->s : p->k) - 1), pl.bin_gran, a.cls_lo, a.cls_hi, a.cls_pretend);  // (the kernels step over k-mers / s-mers)
        if (brc != BSK_OK) return brc;
        a.desc = b->bdesc;
        a.rflags = b->rflags ? b->bflags : nullptr;
        a.len_mask = 0xfffu;
        a.binned = 1;
    }
    const bool lists = pl.which == K_SYN_PK || pl.which == K_MIN_PK || pl.which == K_MIN_RING || pl.which == K_SYN_SEL || pl.which == K_MIN_PKD;
    const u64 fixcap = lists ? syn_pk_fixcap(b->n, pl.grid) : 0;  // u32 entries, behind one u32 count per workgroup
    int rc = ensure_scratch(ctx, std::max<u32>(lists ? (u32)((fixcap + (u64)pl.grid) / 2 + 2) : pl.slab ? 1 : pl.nunits, pl.mixed ? pl.side_nunits : 0), pl.ring_entries);
    if (rc != BSK_OK) return rc;
    a.lookback = ctx->d_lookback;
    a.fixlist = ctx->d_lookback;
    a.fixcap = (u32)fixcap;
    a.list_grid = (u32)pl.grid;
    a.rlist = reinterpret_cast<u32 *>(ctx->d_lookback);  // (the slab kernels use no look-back words: the list of reads lives there)
    a.unit_rows = (u32)(pl.slab_unit / 64);
    if (pl.which == K_MIN_PK || pl.which == K_MIN_RING) {  // slab of a listed read (k_minimizer_dense<W, true>): one tuple per window, whole 128-byte lines
        const u64 nwin_max = b->maxlen + 2 > (u32)(p->k + p->w) ? (u64)b->maxlen - p->k - p->w + 2 : 1;
        a.slab_read = (nwin_max + 15) & ~(u64)15;
    }
    if (pl.which == K_MIN_PKD) {  // (the main kernel has per-read slabs of its own: KArgs::slab_read; the list pass runs with list_slab)
        const u64 nwin_max = b->maxlen + 2 > (u32)(p->k + p->w) ? (u64)b->maxlen - p->k - p->w + 2 : 1;
        a.list_slab = (nwin_max + 15) & ~(u64)15;
    }
    a.ring_h = ctx->d_ring_h;
    a.ring_p = ctx->d_ring_p;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_ticket, 0, 8 * sizeof(u32), ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_total, 0, 4 * sizeof(u64), ctx->stream));
    if (!pl.slab) HIPCHK(ctx, hipMemsetAsync(ctx->d_lookback, 0, (size_t)pl.nunits * sizeof(u64), ctx->stream));
    if (ev0 && !cs) HIPCHK(ctx, hipEventRecord(ev0, ctx->stream));
    // The ASCII side launch of a mixed batch (a general per-lane kernel over the reads with a non-ACGT letter: 60-100 Gbases/s on a grid
    // of its own).  Position kinds: it runs BESIDE the main kernel, on the side context's stream and look-back scratch, queued ahead of it,
    // into reference words and status bytes of its own that k_adopt_side copies over the main kernel's afterwards -- behind the main
    // kernel it cost 35 % of the call with 1 % of the reads flagged (1.5 10^9 bases of 150-base reads: 755 against 1 150 Gbases/s).
    // Stream kinds overwrite the runs the main kernel laid out (inplace) and stay behind it.
    auto side_launch = [&](hipStream_t st, u64 *lookback, u64 *refs_to, u8 *status_to) -> int {
        KArgs sd = a;
        sd.desc = b->desc;  // (the side launch names its reads by their batch positions)
        sd.rflags = b->rflags;
        sd.len_mask = 0xffffffu;
        sd.binned = 0;
        sd.cls_lo = sd.cls_hi = sd.cls_pretend = 0;
        sd.subset = b->subset;
        sd.nsub = b->nsub;
        sd.nunits = pl.side_nunits;
        sd.out_base = res->main_cap;
        sd.cap = res->cap;
        sd.uniform_len = 0;
        sd.inplace = !kind_has_pos(p->kind);  // stream kinds: overwrite the read's own run, keep the layout contiguous
        sd.ticket = ctx->d_ticket + 2;
        sd.total = ctx->d_total + 2;
        sd.ring_w = pl.side_ring_w;
        sd.lookback = lookback;
        sd.refs = refs_to;
        sd.status = status_to;
        HIPCHK(ctx, hipMemsetAsync(lookback, 0, (size_t)pl.side_nunits * sizeof(u64), st));
        switch (pl.side_which) {
            case K_MIN_GEN_A: hipLaunchKernelGGL(k_minimizer_generic<1>, dim3(pl.side_grid), dim3(64), 0, st, sd); break;
            case K_MIN_DENSE_A:
                sd.slab_read = pl.side_slab;
                dense_minimizer_ascii_launch(p->w, pl.side_grid, st, sd);
                break;
            case K_SYN_FAST_A:
                sd.ovf_base = res->main_cap + (u64)pl.side_nunits * 64 * pl.side_slab;
                sd.ovf_cap = res->cap > sd.ovf_base ? res->cap - sd.ovf_base : 0;
                fast_syncmer_ascii_launch(p->k - p->s, pl.side_grid, st, sd);
                break;
            case K_NT_A: hipLaunchKernelGGL(k_nthash_stream<1>, dim3(pl.side_grid), dim3(64), 0, st, sd); break;
            case K_SYN_A: hipLaunchKernelGGL(k_syncmer<1>, dim3(pl.side_grid), dim3(64), 0, st, sd); break;
            case K_KMER_A: hipLaunchKernelGGL(k_kmer<1>, dim3(pl.side_grid), dim3(64), 0, st, sd); break;
            case K_SIM_A: hipLaunchKernelGGL(k_simhash<1>, dim3(pl.side_grid), dim3(64), 0, st, sd); break;
            default: ctx->err = "mixed plan without an ASCII kernel"; return BSK_ERR_DEVICE;
        }
        return BSK_OK;
    };
    bool side_early = false;
    if (pl.mixed && kind_has_pos(p->kind) && !ctx->opt.no_side_early && side_ctx(ctx)) {
        bsk_ctx *sc = ctx->side;
        auto grow = [&](int slot, size_t bytes) -> hipError_t {
            if (ctx->tmp_cap[slot] >= bytes) return hipSuccess;
            (void)hipFree(ctx->tmp[slot]);
            ctx->tmp[slot] = nullptr;
            ctx->tmp_cap[slot] = 0;
            const hipError_t e = hipMalloc(&ctx->tmp[slot], bytes + bytes / 4 + 256);
            if (e == hipSuccess) ctx->tmp_cap[slot] = bytes + bytes / 4 + 256;
            return e;
        };
        if (ensure_scratch(sc, pl.side_nunits, 0) == BSK_OK && grow(28, (size_t)b->n * 8) == hipSuccess && grow(29, (size_t)b->n) == hipSuccess) {
            // (behind the counters' memsets above: the side kernel's ticket and total live beside the main kernel's)
            HIPCHK(ctx, hipEventRecord(ctx->ev_mix0, ctx->stream));
            HIPCHK(ctx, hipStreamWaitEvent(sc->stream, ctx->ev_mix0, 0));
            const int src = side_launch(sc->stream, sc->d_lookback, (u64 *)ctx->tmp[28], (u8 *)ctx->tmp[29]);
            if (src != BSK_OK) return src;
            side_early = true;
        } else {
            (void)hipGetLastError();
        }
    }
    // every read of the batch is the side launch's (a non-ACGT letter in each): nothing of the main kernel's would be kept
    const bool main_moot = side_early && !cs && b->nsub == b->n && (pl.side_which == K_MIN_DENSE_A || pl.side_which == K_SYN_FAST_A);
    if (!main_moot) switch (pl.which) {
        case K_MIN_GEN_P: hipLaunchKernelGGL(k_minimizer_generic<0>, dim3(pl.grid), dim3(64), 0, ctx->stream, a); break;
        case K_MIN_GEN_A: hipLaunchKernelGGL(k_minimizer_generic<1>, dim3(pl.grid), dim3(64), 0, ctx->stream, a); break;
        case K_NT_P: hipLaunchKernelGGL(k_nthash_stream<0>, dim3(pl.grid), dim3(64), 0, ctx->stream, a); break;
        case K_NT_A: hipLaunchKernelGGL(k_nthash_stream<1>, dim3(pl.grid), dim3(64), 0, ctx->stream, a); break;
        case K_MIN_FAST: fast_minimizer_launch(pl.fast_w, pl.grid, ctx->stream, a); break;
        case K_MIN_PK: pk_minimizer_launch(pl.fast_w, b->maxlen > pk_minimizer_short_bases(), pl.grid, ctx->stream, a); break;
        case K_MIN_RING: ring_minimizer_launch(pl.fast_w, b->maxlen > ring_minimizer_short_bases(), pl.grid, ctx->stream, a); break;
        case K_MIN_DENSE: dense_minimizer_launch(pl.fast_w, pl.grid, ctx->stream, a); break;
        case K_MIN_PKD: pkd_minimizer_launch(pl.fast_w, pl.grid, ctx->stream, a); break;
        case K_MIN_DENSE_A:
        case K_SYN_FAST_A: break;  // (side launches' kernels only)
#ifdef BSK_EXPERIMENTS
        case K_MIN_SEG: seg_minimizer_launch(pl.fast_w, pl.grid, ctx->stream, a); break;
        case K_MIN_WPR: wpr_minimizer_launch(pl.grid, ctx->stream, a); break;
#else
        case K_MIN_SEG:
        case K_MIN_WPR: break;
#endif
        case K_SYN_P: hipLaunchKernelGGL(k_syncmer<0>, dim3(pl.grid), dim3(64), 0, ctx->stream, a); break;
        case K_SYN_A: hipLaunchKernelGGL(k_syncmer<1>, dim3(pl.grid), dim3(64), 0, ctx->stream, a); break;
        case K_KMER_P: hipLaunchKernelGGL(k_kmer<0>, dim3(pl.grid), dim3(64), 0, ctx->stream, a); break;
        case K_KMER_A: hipLaunchKernelGGL(k_kmer<1>, dim3(pl.grid), dim3(64), 0, ctx->stream, a); break;
        case K_SIM_P: hipLaunchKernelGGL(k_simhash<0>, dim3(pl.grid), dim3(64), 0, ctx->stream, a); break;
        case K_SIM_A: hipLaunchKernelGGL(k_simhash<1>, dim3(pl.grid), dim3(64), 0, ctx->stream, a); break;
        case K_PROT_HASH: hipLaunchKernelGGL(k_prot_hash, dim3(pl.grid), dim3(64), 0, ctx->stream, a); break;
        case K_PROT_MIN: hipLaunchKernelGGL(k_prot_minimizer, dim3(pl.grid), dim3(64), 0, ctx->stream, a); break;
        case K_SYN_FAST: fast_syncmer_launch(pl.fast_w, pl.grid, ctx->stream, a); break;
        case K_SYN_PK:
            if (pl.syn_fused) pf_syncmer_launch(pl.fast_w, pl.syn_long, pl.grid, std::min(pl.grid, ctx->cus * 8), ctx->stream, a);
            else pk_syncmer_launch(pl.fast_w, pl.syn_long, pl.grid, std::min(pl.grid, ctx->cus * 8), ctx->stream, a);
            break;
#ifndef BSK_EXPERIMENTS
        case K_SYN_SEL: break;
#else
        case K_SYN_SEL: {
            // scratch of the two passes (context pool, grow-only): selection words [unit][nb][64], per read offset | count, per unit total / base
            const u32 ns_max = b->maxlen + 1 > (u32)p->s ? b->maxlen - (u32)p->s + 1 : 1;
            const u32 nb = (ns_max + (u32)pl.fast_w - 1) / (u32)pl.fast_w;  // fused blocks: i0 = W, 2W, ... < ns_max (one spare)
            const u32 nblocks = (pl.nunits + 1023u) / 1024u;
            auto pool = [&](int slot, size_t bytes, void **outp) -> hipError_t {
                if (ctx->tmp_cap[slot] < bytes) {
                    (void)hipFree(ctx->tmp[slot]);
                    ctx->tmp[slot] = nullptr;
                    ctx->tmp_cap[slot] = 0;
                    const size_t want = bytes + bytes / 4 + 256;
                    const hipError_t e = hipMalloc(&ctx->tmp[slot], want);
                    if (e != hipSuccess) return e;
                    ctx->tmp_cap[slot] = want;
                }
                *outp = ctx->tmp[slot];
                return hipSuccess;
            };
            HIPCHK(ctx, pool(24, (size_t)pl.nunits * nb * 64 * 4, (void **)&a.sel_mask));
            HIPCHK(ctx, pool(25, (size_t)pl.nunits * 64 * 4, (void **)&a.sel_cnt));
            HIPCHK(ctx, pool(26, (size_t)pl.nunits * 4 + 64, (void **)&a.sel_utot));
            HIPCHK(ctx, pool(27, ((size_t)pl.nunits + nblocks + 8) * 8, (void **)&a.sel_ubase));
            a.sel_lookback = a.sel_ubase + pl.nunits;
            a.sel_nb = nb;
            HIPCHK(ctx, hipMemsetAsync(a.sel_lookback, 0, (size_t)nblocks * 8, ctx->stream));
            sel_syncmer_launch(pl.fast_w, pl.grid, std::min(pl.grid, ctx->cus * 8), ctx->cus, b->maxlen / 16 + 6, ctx->stream, a);  // (words: the last k-mer's five words start at word (L - k) / 16; pad_words covers the overrun)
            break;
        }
#endif
        case K_PROT_MIN_FAST:
            if (pl.fused_dna) {
                a.frame = p->frame;
                a.lut = ctx->d_lut;
                fast_prot_dna_launch(pl.fast_w, pl.fast_k, pl.grid, ctx->stream, a);
            } else {
                fast_prot_launch(pl.fast_w, pl.fast_k, pl.grid, ctx->stream, a);
            }
            break;
        case K_PROT_HASH_FAST:
            if (pl.fused_dna) {
                a.frame = p->frame;
                a.lut = ctx->d_lut;
                fast_prot_hash_dna_launch(pl.fast_k, pl.grid, ctx->stream, a);
            } else {
                fast_prot_hash_launch(pl.fast_k, pl.grid, ctx->stream, a);
            }
            break;
        case K_SIM_FAST:
            if (pl.fast_k == 1) {
                if (pl.fast_w == 5) hipLaunchKernelGGL((k_simhash_fast<5, BSK_SIM_SHORT_WORDS>), dim3(pl.grid), dim3(64), 0, ctx->stream, a);
                else hipLaunchKernelGGL((k_simhash_fast<6, BSK_SIM_SHORT_WORDS>), dim3(pl.grid), dim3(64), 0, ctx->stream, a);
            } else if (pl.fast_k == 2) {
                if (pl.fast_w == 5) hipLaunchKernelGGL((k_simhash_fast<5, BSK_SIM_MID_WORDS>), dim3(pl.grid), dim3(64), 0, ctx->stream, a);
                else hipLaunchKernelGGL((k_simhash_fast<6, BSK_SIM_MID_WORDS>), dim3(pl.grid), dim3(64), 0, ctx->stream, a);
            } else if (pl.fast_w == 5) hipLaunchKernelGGL(k_simhash_fast<5>, dim3(pl.grid), dim3(64), 0, ctx->stream, a);
            else hipLaunchKernelGGL(k_simhash_fast<6>, dim3(pl.grid), dim3(64), 0, ctx->stream, a);
            break;
        case K_NT_FAST:
#ifdef BSK_EXPERIMENTS
            if (pl.compact) {
                if (a.kind == BSK_KMER) hipLaunchKernelGGL((k_nthash_fast<2, true>), dim3(pl.grid), dim3(64), 0, ctx->stream, a);
                else if (a.canonical) hipLaunchKernelGGL((k_nthash_fast<1, true>), dim3(pl.grid), dim3(64), 0, ctx->stream, a);
                else hipLaunchKernelGGL((k_nthash_fast<0, true>), dim3(pl.grid), dim3(64), 0, ctx->stream, a);
            } else
#endif
            if (a.kind == BSK_KMER && a.one_strand) hipLaunchKernelGGL(k_nthash_fast<4>, dim3(pl.grid), dim3(64), 0, ctx->stream, a);
            else if (a.kind == BSK_KMER && !a.canonical) hipLaunchKernelGGL(k_nthash_fast<3>, dim3(pl.grid), dim3(64), 0, ctx->stream, a);
            else if (a.kind == BSK_KMER) hipLaunchKernelGGL(k_nthash_fast<2>, dim3(pl.grid), dim3(64), 0, ctx->stream, a);
            else if (a.canonical) hipLaunchKernelGGL(k_nthash_fast<1>, dim3(pl.grid), dim3(64), 0, ctx->stream, a);
            else hipLaunchKernelGGL(k_nthash_fast<0>, dim3(pl.grid), dim3(64), 0, ctx->stream, a);
            break;
    }
    if (cs) {  // the other classes' reads: their reference words point into the tail (before the ASCII side launch, which owns the reads with an N)
        HIPCHK(ctx, hipEventRecord(ctx->ev_side_done, ctx->side->stream));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_side_done, 0));
        const int arc = adopt_parts(ctx, cs, res);
        if (arc != BSK_OK) return arc;
    }
    if (pl.mixed && !side_early) {  // the reads with a non-ACGT letter again, from their ASCII bytes, into [main_cap, cap)
        const int src = side_launch(ctx->stream, ctx->d_lookback, res->refs, res->status);
        if (src != BSK_OK) return src;
    }
    if (side_early) {  // ... or it ran beside the main kernel (below): its reference words and status bytes replace the main kernel's
        HIPCHK(ctx, hipEventRecord(ctx->ev_mix1, ctx->side->stream));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_mix1, 0));
        hipLaunchKernelGGL(k_adopt_side, dim3(grid_for(ctx, b->nsub, 256)), dim3(256), 0, ctx->stream, b->subset, (u64)b->nsub, (const u64 *)ctx->tmp[28], (const u8 *)ctx->tmp[29],
                           res->refs, res->status);
    }
    if (ev1) HIPCHK(ctx, hipEventRecord(ev1, ctx->stream));
    HIPCHK(ctx, hipGetLastError());
    res->unit_rows = pl.which == K_MIN_RING;  // what actually ran last (sets.hip picks its gather's shape on it, not on the plan string)
    return BSK_OK;
}

// capacity guess (tuples) for the dense kernels; an undershoot is detected on device and the call re-runs
// with the exact size
u64 estimate_cap(const bsk_batch *b, const bsk_params *p, int circ_ext) {
    if (p->kind == BSK_PROT_HASH && b->alphabet == BSK_ALPHA_DNA) return estimate_cap_n(p, b->n_bases / 3 + b->n, b->n);  // fused: residues
    return estimate_cap_n(p, b->n_bases + b->n * (u64)circ_ext, b->n);
}
u64 estimate_cap_n(const bsk_params *p, u64 bases, u64 nreads) {
    switch (p->kind) {
        case BSK_MINIMIZER:
        case BSK_PROT_MINIMIZER: {
            if (p->w <= 1) return bases + 64;
            double d = PlannerTable::slab_sel_num / (p->w + 1.0);
            if (d > 1.0) d = 1.0;
            return (u64)(bases * d) + nreads + 1024;
        }
        case BSK_SYNCMER: {
            if (p->s == p->k) return bases + 64;
            double d = PlannerTable::slab_sel_num / (p->k - p->s + 1.0);
            if (d > 1.0) d = 1.0;
            return (u64)(bases * d) + nreads + 1024;
        }
        case BSK_NTHASH: return bases + 16 * nreads + 64;  // runs are padded to whole 128-byte lines
        case BSK_KMER: return (p->canonical ? 1 : 2) * bases + 16 * nreads + 64;
        case BSK_PROT_HASH:
        case BSK_SIMHASH: return bases + 16 * nreads + 64;
        default: return bases + 64;
    }
}


// Plan, size, launch (and optionally time) the kernel of p->kind over a prepared batch.
int run_planned(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p, int circ_ext, bsk_result **result, int warmup, int iters,
                       float *kernel_ms) {
    int rc = BSK_OK;
    if (!b->desc && b->alphabet == BSK_ALPHA_DNA) {
        ctx->err = "sequences of 2^24 bases or more are only supported by the kinds that tile (not: two-strand k-mer codes)";
        return BSK_ERR_UNSUPPORTED;
    }
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    auto cleanup = [&](int code) {
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        return code;
    };
    Plan pl;
    rc = make_plan(ctx, b, p, pl);
    if (rc != BSK_OK) return cleanup(rc);
    u64 ovf_cap = pl.slab ? std::max<u64>(65536, pl.slab_total / (pl.which == K_SYN_SEL ? 8 : 50)) : 0;  // (two-pass syncmers: the listed reads' tuples, a few per cent of a DENSE region)
    if (pl.which == K_MIN_PK || pl.which == K_MIN_RING || pl.which == K_MIN_PKD) {
        // the list pass gives every listed read a slab of one tuple per window out of this region (a wavefront claims 64 of them): room
        // for 1.5 % of the reads -- low-complexity tails are per cent of real reads -- before the call has to be sized again
        const u64 nwin_max = b->maxlen + 2 > (u32)(p->k + p->w) ? (u64)b->maxlen - p->k - p->w + 2 : 1;
        // (the unit-row kernel lists the reads whose lane outran the ring as well: 1.6 / 2.6-3.1 / 3.0 % of the reads at 200 / 250 / 300 bases,
        // w = 11 -- with room for 1.5 % every first call on such a batch ran the kernel twice; room for 4.5 % there)
        const u64 list_units = pl.which == K_MIN_RING ? 3 * (b->n / 64) + 64 : b->n / 64 + 64;
        ovf_cap += list_units * ((nwin_max + 15) & ~(u64)15);
    }
    if (pl.slab && *result && (*result)->ovf_cap > ovf_cap) ovf_cap = (*result)->ovf_cap;
    if (pl.slab && *result && ctx->in_resize) ovf_cap = std::max(ovf_cap, 2 * (*result)->ovf_cap + 65536);  // a timed re-run outgrew the region: twice the room
    u64 cap = pl.slab ? pl.slab_total + ovf_cap : estimate_cap(b, p, circ_ext);
    const u32 side_len = std::max(b->maxlen, b->side_maxlen);  // a class view's ASCII side launch covers the flagged reads of EVERY class, not only the bulk's
    u64 side_cap = (pl.mixed && kind_has_pos(p->kind)) ? estimate_cap_n(p, b->nsub * (u64)side_len, b->nsub) : 0;  // maxlen already includes a circular extension
    if (pl.mixed && pl.side_which == K_MIN_DENSE_A) side_cap = (u64)pl.side_nunits * 64 * pl.side_slab + 64;
    if (pl.mixed && pl.side_which == K_SYN_FAST_A) side_cap += (u64)pl.side_nunits * 64 * pl.side_slab + 64;  // (unit slabs, then the dense estimate above as their overflow region)
    if (*result && pl.mixed && (*result)->main_cap && (*result)->cap > (*result)->main_cap) {
        cap = std::max(cap, (*result)->main_cap);
        side_cap = std::max(side_cap, (*result)->cap - (*result)->main_cap);
    } else if (*result && !pl.mixed && (*result)->cap > cap) {
        cap = (*result)->cap;
    }
    // bsk_sketch always runs (and sizes) once; bsk_sketch_timed on an existing result only repeats the launch -- of the plan the result
    // was sized for, on the batch it was sized for (anything else could write past `cap`)
    const bool sizing = *result == nullptr || warmup + iters == 0;
    if (!sizing && !plan_recall(*result, b, p, circ_ext, pl)) {
        ctx->err = "bsk_sketch_timed: the result was not sized for this batch and these parameters: call bsk_sketch first";
        return cleanup(BSK_ERR_ARG);
    }
    bool side_fell_back = false, ovf_grown = false;
    struct SideGuard {
        bsk_ctx *c;
        ~SideGuard() { c->no_side_fast = false; }
    } side_guard{ctx};
    for (int attempt = 0; sizing && attempt < 3; ++attempt) {
        rc = result_prepare(ctx, result, b->n, p->kind, cap + side_cap, (ctx->cls && b == ctx->cls->view) ? ctx->cls->tail : 0);
        if (rc == BSK_ERR_NOMEM && (pl.which == K_MIN_DENSE || pl.which == K_MIN_PKD || pl.which == K_MIN_SEG || pl.which == K_MIN_WPR || pl.which == K_PROT_MIN_FAST) && attempt < 2) {
            // per-read slabs did not fit the device: the unit-slab / dense-CSR kernels need far less
            ctx->no_prot_fast = true;
            ctx->no_dense = true;
            pl = Plan();
            rc = make_plan(ctx, b, p, pl);
            ctx->no_prot_fast = false;
            ctx->no_dense = false;
            if (rc != BSK_OK) return cleanup(rc);
            cap = pl.slab ? pl.slab_total + std::max<u64>(65536, pl.slab_total / 50) : estimate_cap(b, p, circ_ext);
            continue;
        }
        if (rc != BSK_OK) return cleanup(rc);
        bsk_result *res = *result;
        res->main_cap = pl.mixed ? cap : 0;
        res->ovf_cap = pl.slab ? (pl.mixed ? cap : res->cap) - pl.slab_total : 0;
        plan_name(pl, p, false, ctx->cus, res);
        rc = launch(ctx, b, p, res, circ_ext, pl, nullptr, nullptr);
        if (rc != BSK_OK) return cleanup(rc);
        if (pl.nunits == 0) {  // empty batch: nothing was launched, the scratch counters are stale
            res->n_tuples = 0;
            if (ctx->defer) HIPCHK(ctx, hipMemsetAsync(ctx->d_ticket + 20, 0, 4 * sizeof(u32), ctx->stream));
            break;
        }
        // (deferred only where what the launch leaves behind is in range WHATEVER it overflowed: slab kernels -- a read's reference word names its
        // own slab -- and the stream kinds, whose counts follow from the lengths.  The dense look-back kernels size by an estimate, and after an
        // undershoot their reference words point past the arrays: the stitch pass would follow them -- the memory fault of fuzz seed 21002744's
        // neighbourhood, round 6.  They take the sizing loop below; the caller's bound-sized tile table serves either way.)
        const bool defer_now = ctx->defer && (pl.slab || !kind_has_pos(p->kind));
        if (ctx->defer && !defer_now) HIPCHK(ctx, hipMemsetAsync(ctx->d_ticket + 20, 0, 4 * sizeof(u32), ctx->stream));
        if (defer_now) {  // nothing is read back: the launch's flags are parked where later passes leave them alone, the caller looks at them
            HIPCHK(ctx, hipMemcpyAsync(ctx->d_ticket + 20, ctx->d_ticket, 4 * sizeof(u32), hipMemcpyDeviceToDevice, ctx->stream));
            res->n_tuples = res->cap;  // (an upper bound; the caller sizes by it)
            plan_record(res, b, p, circ_ext, pl);
            return cleanup(BSK_OK);
        }
        hipError_t e = hipMemcpyAsync(ctx->h_pinned, ctx->d_total, 4 * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(ctx->h_pinned + 4, ctx->d_ticket, 4 * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream);
        const bool has_parts = ctx->cls && b == ctx->cls->view;
        if (e == hipSuccess && has_parts) e = hipMemcpyAsync(ctx->h_pinned + 6, ctx->d_ticket + 16, sizeof(u32), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return cleanup(fail_hip(ctx, e, "bsk_sketch run"));
        if (has_parts && (*(const u32 *)(ctx->h_pinned + 6) || ((ctx->opt.test_overflow & 2u) && ctx->cls_round == 0 && !ctx->part_grow)))
            return cleanup(BSK_REPLAN_CLASS);  // (the parts were sized by launches of their own; this one used more)
        const u64 total = ctx->h_pinned[0], ovf_used = ctx->h_pinned[1], side_end = ctx->h_pinned[2];
        const u32 ovf = ((u32 *)(ctx->h_pinned + 4))[1], side_ovf = ((u32 *)(ctx->h_pinned + 4))[3];
        res->n_tuples = total;
        if (ctx->opt.timing) fprintf(stderr, "[bsk] sizing attempt %d: overflow region %llu of %llu tuples used, flags %u / %u\n", attempt, (unsigned long long)ovf_used, (unsigned long long)res->ovf_cap, ovf, side_ovf);
        if (!ovf && !side_ovf) {
            // The overflow region's use varies from launch to launch by a few slabs (the list pass takes 64 slabs per wavefront and segment, and
            // which workgroup lists which reads follows the tickets): a launch that fitted by less than a fifth is sized again with room to
            // spare, or a timed re-run of the same plan overflows now and then (6 10^7 x 250 bases on k_minimizer_ring: 277.07-277.33 M tuples
            // used of 277.21 M -- two of six bench runs failed).
            if (pl.slab && res->ovf_cap && ovf_used * 5 > res->ovf_cap * 4 && !ovf_grown && attempt < 2) {
                ovf_grown = true;
                cap = pl.slab_total + ovf_used + ovf_used / 4 + 65536;
                continue;
            }
            break;
        }
        if (side_ovf && (pl.side_which == K_SYN_FAST_A || pl.side_which == K_MIN_DENSE_A) && !side_fell_back) {  // the staged side kernels' regions are sized up front: plan again with the general one
            ctx->no_side_fast = true;  // (for the rest of this call: side_guard)
            pl = Plan();
            rc = make_plan(ctx, b, p, pl);
            if (rc != BSK_OK) return cleanup(rc);
            cap = pl.slab ? pl.slab_total + ovf_cap : estimate_cap(b, p, circ_ext);
            side_cap = (pl.mixed && kind_has_pos(p->kind)) ? estimate_cap_n(p, b->nsub * (u64)side_len, b->nsub) : 0;
            --attempt;  // (the general kernel keeps its own two tries: an estimate, then the exact size -- fuzz seed 11003764: k = 21, s = 1)
            side_fell_back = true;
            continue;
        }
        if (side_ovf && attempt < 2) side_cap = side_end - res->main_cap + 64;  // dense side kernel: its end is exact even when it overflowed
        if (!ovf && attempt < 2) continue;
        if (attempt == 2) {
            ctx->err = "result capacity overflow after exact re-size";
            return cleanup(BSK_ERR_DEVICE);
        }
        if (pl.which == K_SYN_SEL && !(ovf & 2u)) {  // the dense region (or the listed reads' region) was too small: total = what pass 2 needs
            ctx->sel_need = total + total / 32 + 4096;
            pl = Plan();
            rc = make_plan(ctx, b, p, pl);
            ctx->sel_need = 0;
            if (rc != BSK_OK) return cleanup(rc);
            ovf_cap = std::max<u64>(ovf_cap, ovf_used + ovf_used / 4 + 65536);
            cap = pl.slab_total + ovf_cap;
            continue;
        }
        if (pl.which == K_PROT_MIN_FAST || pl.which == K_MIN_DENSE || ((pl.which == K_SYN_PK || pl.which == K_MIN_PK || pl.which == K_MIN_RING || pl.which == K_SYN_SEL || pl.which == K_MIN_PKD) && (ovf & 2u))) {  // a sequence outgrew its slab (unusual density), or too many reads with key ties: re-plan without that kernel
            ctx->no_prot_fast = true;
            ctx->no_dense = true;
            ctx->no_syn_pk = true;
            pl = Plan();  // not just `which`: the slab fields of the abandoned plan must go too (they size the look-back scratch)
            rc = make_plan(ctx, b, p, pl);
            ctx->no_prot_fast = false;
            ctx->no_dense = false;
            ctx->no_syn_pk = false;
            if (rc != BSK_OK) return cleanup(rc);
            cap = pl.slab ? pl.slab_total + std::max<u64>(65536, pl.slab_total / 50) : estimate_cap(b, p, circ_ext);
            continue;
        }
        cap = pl.slab ? pl.slab_total + ovf_used + ovf_used / 4 + 65536 : total + 64;  // size known now: re-run once
    }
    if (sizing && *result) plan_record(*result, b, p, circ_ext, pl);
    if (sizing && (pl.mixed || pl.slab || pl.which == K_NT_FAST || pl.which == K_PROT_HASH_FAST || pl.which == K_SIM_FAST || (ctx->cls && b == ctx->cls->view)) && b->n) {  // slab / line-padded kernels: sum the per-read counts once
        HIPCHK(ctx, hipMemsetAsync(ctx->d_total, 0, sizeof(u64), ctx->stream));
        hipLaunchKernelGGL(k_sum_counts, dim3(grid_for(ctx, b->n, 256)), dim3(256), 0, ctx->stream, (*result)->refs, b->n, ctx->d_total);
        hipError_t e = hipMemcpyAsync(ctx->h_pinned, ctx->d_total, sizeof(u64), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return cleanup(fail_hip(ctx, e, "bsk_sketch count"));
        (*result)->n_tuples = ctx->h_pinned[0];
    }
    // timed repetitions (same result buffers; capacity is now known to be sufficient).  All launches are queued
    // back to back; the per-kernel HIP events are read after one final stream synchronisation.
    std::vector<hipEvent_t> evs;
    if (kernel_ms)
        for (int i = 0; i < 2 * iters; ++i) {
            hipEvent_t e = nullptr;
            (void)hipEventCreate(&e);
            evs.push_back(e);
        }
    auto drop_events = [&]() {
        for (hipEvent_t e : evs)
            if (e) (void)hipEventDestroy(e);
    };
    if (*result) plan_name(pl, p, false, ctx->cus, *result);
    for (int it = 0; it < warmup + iters; ++it) {
        const bool timed = it >= warmup && kernel_ms;
        rc = launch(ctx, b, p, *result, circ_ext, pl, timed ? evs[2 * (it - warmup)] : nullptr,
                    timed ? evs[2 * (it - warmup) + 1] : nullptr);
        if (rc != BSK_OK) {
            drop_events();
            return cleanup(rc);
        }
    }
    if (warmup + iters > 0) {
        hipError_t e = hipMemcpyAsync(ctx->h_pinned + 2, ctx->d_ticket, 4 * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream);
        const bool has_parts = ctx->cls && b == ctx->cls->view;
        if (e == hipSuccess && has_parts) e = hipMemcpyAsync(ctx->h_pinned + 6, ctx->d_ticket + 16, sizeof(u32), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess && has_parts && (*(const u32 *)(ctx->h_pinned + 6) || ((ctx->opt.test_overflow & 4u) && !ctx->in_resize))) {  // (only the LAST launch's flags are left: enough to know the sizes no longer hold)
            drop_events();
            if (!ctx->in_resize) return cleanup(BSK_REPLAN_CLASS);
            ctx->err = "class plan: a part outgrew its slabs again after it was sized with room: call bsk_sketch first";
            return cleanup(BSK_ERR_ARG);
        }
        if (e == hipSuccess && ((((u32 *)(ctx->h_pinned + 2))[1] | ((u32 *)(ctx->h_pinned + 2))[3]) || ((ctx->opt.test_overflow & 1u) && !ctx->in_resize && pl.slab && !pl.mixed))) {
            drop_events();
            if (!ctx->in_resize && pl.slab && !pl.mixed) return cleanup(BSK_RESIZE);  // the overflow region's use varies by a few slabs per launch: size again with room, once
            {
                char msg[160];
                snprintf(msg, sizeof msg, "result too small for this batch (overflow flags %u / side %u: 1 = a region or slab, 2 = a list segment): call bsk_sketch first",
                         ((u32 *)(ctx->h_pinned + 2))[1], ((u32 *)(ctx->h_pinned + 2))[3]);
                ctx->err = msg;
            }
            return cleanup(BSK_ERR_ARG);
        }
        for (int i = 0; e == hipSuccess && kernel_ms && i < iters; ++i) e = hipEventElapsedTime(&kernel_ms[i], evs[2 * i], evs[2 * i + 1]);
        drop_events();
        if (e != hipSuccess) return cleanup(fail_hip(ctx, e, "bsk_sketch_timed sync"));
    }
    return cleanup(BSK_OK);
}

// run_planned for callers that time an existing result: a launch that outgrows the regions the result was sized with (their use varies
// by a few slabs from launch to launch: which workgroup lists which reads follows the tickets) sizes the result again with twice the room
// and repeats the timed launches -- once; a second overflow is the caller's error as before.
int run_planned_resizing(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p, int circ_ext, bsk_result **result, int warmup, int iters,
                                float *kernel_ms) {
    int rc = run_planned(ctx, b, p, circ_ext, result, warmup, iters, kernel_ms);
    if (rc != BSK_RESIZE) return rc;
    ctx->in_resize = true;
    rc = run_planned(ctx, b, p, circ_ext, result, 0, 0, nullptr);
    if (rc == BSK_OK) rc = run_planned(ctx, b, p, circ_ext, result, warmup, iters, kernel_ms);
    ctx->in_resize = false;
    return rc;
}


// workgroups per CU of the kernels this translation unit instantiates (make_plan_enc asks by name)
int occ(OccId id) {
    switch (id) {
        case OCC_MIN_GEN_P: return blocks_per_cu(k_minimizer_generic<0>);
        case OCC_MIN_GEN_A: return blocks_per_cu(k_minimizer_generic<1>);
        case OCC_NT_FAST0: return blocks_per_cu(k_nthash_fast<0>);
        case OCC_NT_FAST1: return blocks_per_cu(k_nthash_fast<1>);
        case OCC_NT_FAST2: return blocks_per_cu(k_nthash_fast<2>);
        case OCC_NT_FAST3: return blocks_per_cu(k_nthash_fast<3>);
        case OCC_NT_FAST4: return blocks_per_cu(k_nthash_fast<4>);
#ifdef BSK_EXPERIMENTS
        case OCC_NT_FAST0C: return blocks_per_cu(k_nthash_fast<0, true>);
        case OCC_NT_FAST1C: return blocks_per_cu(k_nthash_fast<1, true>);
        case OCC_NT_FAST2C: return blocks_per_cu(k_nthash_fast<2, true>);
#else
        case OCC_NT_FAST0C:
        case OCC_NT_FAST1C:
        case OCC_NT_FAST2C: return 1;
#endif
        case OCC_NT_P: return blocks_per_cu(k_nthash_stream<0>);
        case OCC_NT_A: return blocks_per_cu(k_nthash_stream<1>);
        case OCC_SYN_P: return blocks_per_cu(k_syncmer<0>);
        case OCC_SYN_A: return blocks_per_cu(k_syncmer<1>);
        case OCC_KMER_P: return blocks_per_cu(k_kmer<0>);
        case OCC_KMER_A: return blocks_per_cu(k_kmer<1>);
        case OCC_SIMF_5S: return blocks_per_cu(k_simhash_fast<5, BSK_SIM_SHORT_WORDS>);
        case OCC_SIMF_6S: return blocks_per_cu(k_simhash_fast<6, BSK_SIM_SHORT_WORDS>);
        case OCC_SIMF_5M: return blocks_per_cu(k_simhash_fast<5, BSK_SIM_MID_WORDS>);
        case OCC_SIMF_6M: return blocks_per_cu(k_simhash_fast<6, BSK_SIM_MID_WORDS>);
        case OCC_SIMF_5: return blocks_per_cu(k_simhash_fast<5>);
        case OCC_SIMF_6: return blocks_per_cu(k_simhash_fast<6>);
        case OCC_SIM_P: return blocks_per_cu(k_simhash<0>);
        case OCC_SIM_A: return blocks_per_cu(k_simhash<1>);
        case OCC_PROT_HASH: return blocks_per_cu(k_prot_hash);
        case OCC_PROT_MIN: return blocks_per_cu(k_prot_minimizer);
    }
    return 1;
}
