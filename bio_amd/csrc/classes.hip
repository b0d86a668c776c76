// classes.hip -- class plans: one plan per LENGTH CLASS of a batch instead of one plan per batch.
#include "host_internal.hpp"
#include "kernels_host.hpp"

// ------------------------------------------------------------------------------------
// class plans: one plan per LENGTH CLASS of a batch instead of one plan per batch
// ------------------------------------------------------------------------------------
// The reference sketches one sequence at a time: a 5-kb contig costs 5 kb, whatever else is in the file (sketch.go:46, :85-94).  A batch
// plan keyed on the longest read does not: one 400-base read moved 10^8 x 150 bases from k_minimizer_pk to k_minimizer_dense, one 5-kb
// read moved them onto tiles.  A class plan cuts the batch by length at the points where the planner's choice changes (LenHist: known on
// the host since the batch was created), runs the BULK class -- the one with most bases -- over a view of the batch in which every other
// read has length 0, and every other class as a batch of its own (its descriptors gathered, the words shared) whose slabs live in the
// TAIL of the parent's arrays; k_adopt_refs then points those reads' reference words there.  Callers see one result.
void class_set_free(ClassSet *cs) {
    if (!cs) return;
    for (auto &pt : cs->parts) {
        if (pt.res) bsk_result_release(pt.res);
        if (pt.sub) bsk_batch_destroy(pt.sub);
    }
    if (cs->view) bsk_batch_destroy(cs->view);
    delete cs;
}
extern "C" int bsk_result_class_plan(const bsk_result *r, int *n_parts, float *build_ms) {
    if (!r) return BSK_ERR_ARG;
    if (n_parts) *n_parts = r->classes ? (int)r->classes->parts.size() : 0;
    if (build_ms) *build_ms = r->classes ? r->classes->build_ms : 0.0f;
    return BSK_OK;
}

// the parts of a class plan into the tail of `res` (called from the parent's launch, before its own kernel)
// The parts run on the side context's stream (plain launches over batches of their own; their slabs are slices of the parent's tail):
// tiled = false: the parts that are one launch each, queued BEFORE the bulk's kernel so that they take their few CU slots first and
// the bulk's persistent waves fill the rest (the parts' latency-bound launches then overlap with the bulk); tiled = true: the parts that
// run over tiles (sketch_tiled: several kernels and host round trips) -- first of all.
int launch_parts(bsk_ctx *ctx, ClassSet *cs, const bsk_params *p, bsk_result *res, bool tiled) {
    bsk_ctx *side = ctx->side;
    for (auto &pt : cs->parts) {
        if (pt.tiled != tiled) continue;
        const u64 base = res->cap + pt.off;
        if (base + pt.extent > res->alloc_cap) {
            ctx->err = "class plan: the parts do not fit the result's tail";
            return BSK_ERR_DEVICE;
        }
        if (pt.tiled) {  // tiles + stitch into a result of its own, then one copy into the tail
            if (!pt.fresh) {
                side->tile_async = pt.async;  // (no synchronisation of its own: the part must not hold the bulk's launch back -- unless it was sized the round-trip way after an overflow)
                side->tile_sync = !pt.async;
                const int trc = sketch_tiled(side, pt.sub, p, 0, &pt.res, 0, 0, nullptr);
                pt.async = side->tile_was_async;
                side->tile_async = side->tile_sync = false;
                if (trc != BSK_OK) {
                    ctx->err = side->err;
                    return trc;
                }
            }
            pt.fresh = false;
            if (pt.async) hipLaunchKernelGGL(k_fold_word, dim3(1), dim3(1), 0, side->stream, side->d_ticket + 24, ctx->d_ticket + 16);  // the part's overflow flags of its last run
            const u64 T = pt.res->n_tuples;  // (the asynchronous path: an upper bound -- the result's capacity)
            if (T > pt.extent) {
                ctx->err = "class plan: a tiled part outgrew its place in the tail";
                return BSK_ERR_DEVICE;
            }
            if (T) {
                HIPCHK(ctx, hipMemcpyAsync(res->hash + base, pt.res->hash, T * 8, hipMemcpyDeviceToDevice, side->stream));
                if (res->pos && pt.res->pos) HIPCHK(ctx, hipMemcpyAsync(res->pos + base, pt.res->pos, T * 4, hipMemcpyDeviceToDevice, side->stream));
            }
            continue;
        }
        bsk_result *cr = pt.res;
        if (!cr->arrays_borrowed) {  // first launch after the part was sized on arrays of its own
            (void)hipFree(cr->hash);
            (void)hipFree(cr->pos);
        }
        cr->hash = res->hash + base;
        cr->pos = res->pos ? res->pos + base : nullptr;
        cr->arrays_borrowed = true;
        cr->cap = cr->alloc_cap = pt.extent;
        Plan cpl;
        if (!plan_recall(cr, pt.sub, p, 0, cpl)) {
            ctx->err = "class plan: a part lost its plan";
            return BSK_ERR_DEVICE;
        }
        const int rc = launch(side, pt.sub, p, cr, 0, cpl, nullptr, nullptr);
        if (rc != BSK_OK) {
            ctx->err = side->err;
            return rc;
        }
        if (cpl.nunits) hipLaunchKernelGGL(k_fold_flags, dim3(1), dim3(1), 0, side->stream, side->d_ticket, ctx->d_ticket + 16);
    }
    return BSK_OK;
}
int adopt_parts(bsk_ctx *ctx, ClassSet *cs, bsk_result *res) {
    for (auto &pt : cs->parts) {
        if (!pt.n) continue;
        if (pt.tiled)
            hipLaunchKernelGGL(k_adopt_wide, dim3(grid_for(ctx, pt.n, 256)), dim3(256), 0, ctx->stream, pt.list, pt.n, pt.res->wfirst, pt.res->wcount, pt.res->status,
                               res->cap + pt.off, res->refs, res->status);
        else
            hipLaunchKernelGGL(k_adopt_refs, dim3(grid_for(ctx, pt.n, 256)), dim3(256), 0, ctx->stream, pt.list, pt.n, pt.res->refs, pt.res->status, res->cap + pt.off,
                               res->refs, res->status);
    }
    HIPCHK(ctx, hipGetLastError());
    return BSK_OK;
}


// ---- class plans: the decision (host, from the batch's length histogram) ------------------------------------------------------------
// what the planner would run over `n` reads of `bases` bases, the longest `hi` (the pure 2-bit plan: reads with an N are the parent's side launch)
static bool class_sig(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p, u64 n, u64 bases, u32 hi, ClassSig &g) {
    bsk_batch t = *b;  // shallow: only the shape is looked at
    t.n = n;
    t.n_bases = bases;
    t.maxlen = hi;
    t.uniform_len = 0;
    t.n_nonacgt = 0;
    t.subset = nullptr;
    t.nsub = 0;
    Plan pl;
    if (make_plan(ctx, &t, p, pl) != BSK_OK) return false;
    g.which = (int)pl.which;
    g.syn_long = pl.syn_long;
    g.syn_fused = pl.syn_fused;
    g.octave = hi > 1024 ? 63 - __builtin_clzll((u64)hi) : 0;  // (long classes also split by octave: per-read slabs are sized by the class's longest read)
    if (hi > tile_min_for(ctx, b, p)) g.which = -2, g.syn_long = g.syn_fused = false, g.octave = 99;  // tile work: one class, whatever its lengths
    return true;
}
// rough kernel rates in Tbases/s (profiles/r04/robustness.jsonl): only their ratios matter -- is splitting worth its passes?
static double class_rate(const ClassSig &g, double meanlen, bool tiled, int kind) {
    // (planner_table.hpp: the rates of profiles/r06/planner_sweep.jsonl)
    if (tiled || g.which == -2) return PlannerTable::rate(kind == BSK_SYNCMER ? "TILED_SYN" : "TILED_MIN", meanlen);
    switch ((Which)g.which) {
        case K_MIN_PK: return PlannerTable::rate("K_MIN_PK", meanlen);
        case K_MIN_RING: return PlannerTable::rate("K_MIN_RING", meanlen);
        case K_MIN_DENSE: return PlannerTable::rate("K_MIN_DENSE", meanlen);
        case K_MIN_PKD: return PlannerTable::rate("K_MIN_PKD", meanlen);
        case K_MIN_FAST: return PlannerTable::rate("K_MIN_FAST", meanlen);
        case K_SYN_PK: return PlannerTable::rate(g.syn_fused ? (g.syn_long ? "K_SYN_PFL" : "K_SYN_PF") : g.syn_long ? "K_SYN_PKL" : "K_SYN_PK", meanlen);
        case K_SYN_FAST: return PlannerTable::rate("K_SYN_FAST", meanlen);
        default: return PlannerTable::rate("OTHER", meanlen);
    }
}
// -> the classes (ascending) and the index of the bulk; false: keep one plan
bool class_decide(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p, int circ_ext, std::vector<ClassCut> &cuts, int &bulk) {
    if (ctx->opt.no_class || ctx->opt.force_generic || !b->hist || !b->desc || b->alias || b->borrowed || b->alphabet != BSK_ALPHA_DNA || circ_ext || p->circular ||
        (p->kind != BSK_MINIMIZER && p->kind != BSK_SYNCMER) || b->n < (u64)ctx->opt.class_min || b->n >= (1ULL << 32) || b->uniform_len || ctx->opt.no_tiles)
        return false;
    if (p->kind == BSK_SYNCMER && p->s == p->k) return false;  // (runs as the w = 1 minimizer over tiles)
    const LenHist &h = *b->hist;
    // quick exit: the shortest and the longest occupied bucket want the same kernel (two probes of the planner, the common case)
    int b0 = -1, b1 = -1;
    for (int i = 0; i < LenHist::NB; ++i)
        if (h.cnt[i]) {
            if (b0 < 0) b0 = i;
            b1 = i;
        }
    if (b0 < 0 || b0 == b1) return false;
    const u32 tmin = tile_min_for(ctx, b, p);
    ClassSig s0, s1;
    if (!class_sig(ctx, b, p, h.cnt[b0], h.bases[b0], h.hi[b0], s0) || !class_sig(ctx, b, p, h.cnt[b1], h.bases[b1], h.hi[b1], s1)) return false;
    if (s0 == s1 && h.hi[b1] <= tmin) return false;
    cuts.clear();
    for (int i = b0; i <= b1; ++i) {
        if (!h.cnt[i]) continue;
        ClassSig g;
        if (i == b0) g = s0;
        else if (i == b1) g = s1;
        else if (!class_sig(ctx, b, p, h.cnt[i], h.bases[i], h.hi[i], g)) return false;
        if (!cuts.empty() && cuts.back().sig == g) {
            cuts.back().hi = h.hi[i];
            cuts.back().n += h.cnt[i];
            cuts.back().bases += h.bases[i];
        } else {
            const u32 lo = cuts.empty() ? 0u : cuts.back().hi + 1;
            cuts.push_back(ClassCut{lo, h.hi[i], h.lo[i], h.cnt[i], h.bases[i], g});
        }
    }
    if (cuts.size() < 2 || cuts.size() > 8) return false;
    bulk = 0;
    for (size_t i = 1; i < cuts.size(); ++i)
        if (cuts[i].bases > cuts[(size_t)bulk].bases) bulk = (int)i;
    if (cuts[(size_t)bulk].hi > tmin) return false;  // the bulk itself is tile work: the tiled path takes the batch as before
    if (ctx->opt.class_force) return true;
    // is it worth the passes?  one plan: everything at the rate of the longest read's kernel
    ClassSig sall;
    if (!class_sig(ctx, b, p, b->n, b->n_bases, b->maxlen, sall)) return false;
    const double single = (double)b->n_bases / (1e12 * class_rate(sall, (double)b->n_bases / (double)b->n, b->maxlen > tmin, p->kind));  // seconds
    double split = (b->odd && !ctx->opt.class_view) ? 20e-6  // (the host's list of odd sequences: no device pass over the batch)
                                                     : (double)b->n * 16.0 / 1.2e12;  // k_class_cut: 16 bytes per read at the ~1.2 TB/s it reaches
    for (const auto &c : cuts) split += (double)c.bases / (1e12 * class_rate(c.sig, c.n ? (double)c.bases / (double)c.n : 0.0, false, p->kind)) + 60e-6;  // + a launch
    return ctx->opt.class_force || split < 0.95 * single;
}


// lists, views and sub-batches of the classes (device passes on the context's stream; the arrays live in the context's pool)
int class_build(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p, const std::vector<ClassCut> &cuts, int bulk, ClassSet *cs) {
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
    u64 n_out = 0;
    for (size_t i = 0; i < cuts.size(); ++i)
        if ((int)i != bulk) n_out += cuts[i].n;
    // no device pass at all when the batch kept the list of its sequences outside the fullest bucket (bsk_batch::odd), the bulk holds that
    // bucket and the bulk's kernel reads its lengths through desc_len(): the lists are picked on the host, the kernel masks by length
    bool masked = false;
    if (b->odd && b->modal_bucket >= 0 && !ctx->opt.class_view) {
        const u32 mlo = b->hist->lo[b->modal_bucket], mhi = b->hist->hi[b->modal_bucket];
        const Which bw = (Which)cuts[(size_t)bulk].sig.which;
        masked = mlo >= cuts[(size_t)bulk].lo && mhi <= cuts[(size_t)bulk].hi && n_out <= b->odd->size() &&
                 (bw == K_MIN_PK || bw == K_MIN_RING || bw == K_MIN_DENSE || bw == K_MIN_PKD || bw == K_MIN_FAST || bw == K_SYN_PK || bw == K_SYN_FAST);
    }
    u32 *lists = nullptr;
    u64 *view = nullptr, *sdesc = nullptr;
    HIPCHK(ctx, pool(21, (n_out + 64) * 4, (void **)&lists));
    if (!masked) HIPCHK(ctx, pool(22, (b->n + 1024 + 64) * 8, (void **)&view));  // (+ a ticket: k_class_cut writes whole tickets)
    HIPCHK(ctx, pool(23, (n_out + 64) * 8, (void **)&sdesc));
    const u32 nblocks = (u32)((b->n + 1023) / 1024);  // k_class_list: a ticket is 16 rows of 64 reads
    int rc = ensure_scratch(ctx, nblocks, 0);
    if (rc != BSK_OK) return rc;
    const ClassCut &bk = cuts[(size_t)bulk];
    const u32 pretend = bk.shortest == bk.hi ? bk.hi : 0u;  // a fixed-length bulk: the other reads pretend its length in the view
    const u32 tmin = tile_min_for(ctx, b, p);
    // the set keeps the part objects of an earlier call into the same result (their allocations), index by index
    const size_t nparts = cuts.size() - 1;
    for (size_t i = nparts; i < cs->parts.size(); ++i) {
        if (cs->parts[i].res) bsk_result_release(cs->parts[i].res);
        if (cs->parts[i].sub) bsk_batch_destroy(cs->parts[i].sub);
    }
    cs->parts.resize(nparts);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    if (e0) (void)hipEventRecord(e0, ctx->stream);
    ClassCuts cc;
    memset(&cc, 0, sizeof cc);
    cc.ncls = (u32)cuts.size();
    cc.bulk = (u32)bulk;
    cc.pretend = pretend;
    u64 at = 0;
    size_t pi = 0;
    for (size_t i = 0; i < cuts.size(); ++i) {
        cc.hi[i] = cuts[i].hi;
        cc.first[i] = (u32)at;
        if ((int)i == bulk) continue;
        ClassPart &pt = cs->parts[pi++];
        const ClassCut &c = cuts[i];
        pt.list = lists + at;
        pt.n = c.n;
        pt.bases = c.bases;
        pt.lo = c.lo;
        pt.hi = c.hi;
        pt.tiled = c.hi > tmin;
        pt.fresh = false;
        if (!pt.sub) pt.sub = new (std::nothrow) bsk_batch();
        if (!pt.sub) return BSK_ERR_NOMEM;
        bsk_batch *sb = pt.sub;
        sb->ctx = ctx;
        sb->alphabet = b->alphabet;
        sb->pairs = b->pairs;
        sb->n = c.n;
        sb->n_bases = c.bases;
        sb->n_words = b->n_words;
        sb->maxlen = c.hi;
        sb->uniform_len = c.shortest == c.hi ? c.hi : 0;
        sb->words = b->words;
        sb->desc = sdesc + at;
        sb->borrowed = true;
        sb->bin_gran = 0;  // (a binned view of an earlier chunk is stale)
        at += c.n;
    }
    if (masked) {  // the lists from the host's list of odd sequences (ascending), one small copy, the descriptors gathered on the device
        // (staged in the context's pinned buffer when it is large enough -- a batch made from host data on this context left it so:
        // 10^6 entries from pageable memory were 0.9 of the cut's 0.97 ms)
        if (b->d_odd && b->odd->size() >= 65536 && n_out) {  // long lists: split on the device (k_odd_split)
            HIPCHK(ctx, hipMemsetAsync(ctx->d_ticket + 8, 0, 8 * sizeof(u32), ctx->stream));  // [8..15] the classes' cursors
            hipLaunchKernelGGL(k_odd_split, dim3(grid_for(ctx, b->odd->size(), 256)), dim3(256), 0, ctx->stream, b->d_odd, (u64)b->odd->size(), cc, (u32)n_out,
                               ctx->d_ticket + 8, b->desc, lists, sdesc);
        } else {
        std::vector<u32> pageable;
        u32 *host = nullptr;
        if (ctx->h_refs && (u64)ctx->h_refs_cap * 8 >= n_out * 4) host = reinterpret_cast<u32 *>(ctx->h_refs);
        else {
            pageable.resize((size_t)n_out);
            host = pageable.data();
        }
        std::vector<u64> fill(cuts.size(), 0);
        for (const u64 e : *b->odd) {
            const u32 L = (u32)e;
            size_t c = 0;
            while (c + 1 < cuts.size() && L > cuts[c].hi) ++c;
            if ((int)c == bulk) continue;  // (a sequence of the bulk outside the fullest bucket)
            const u64 at_c = (u64)cc.first[c] + fill[c]++;
            if (at_c >= n_out) return fail_arg(ctx, "class plan: the batch's length histogram and its list of odd sequences disagree");
            host[(size_t)at_c] = (u32)(e >> 32);
        }
        for (size_t c = 0; c < cuts.size(); ++c)
            if ((int)c != bulk && fill[c] != cuts[c].n) return fail_arg(ctx, "class plan: the batch's length histogram and its list of odd sequences disagree");
        if (n_out) {
            HIPCHK(ctx, hipMemcpyAsync(lists, host, (size_t)n_out * 4, hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // (the staging buffer is the context's, or goes out of scope)
            hipLaunchKernelGGL(k_gather_desc, dim3(grid_for(ctx, n_out, 256)), dim3(256), 0, ctx->stream, b->desc, lists, n_out, sdesc);
        }
        }
    } else {
        HIPCHK(ctx, hipMemsetAsync(ctx->d_ticket, 0, 16 * sizeof(u32), ctx->stream));  // [8..15] the classes' cursors
        // (the pass is latency-bound per ticket: every wave the CUs hold)
        hipLaunchKernelGGL(k_class_cut, dim3(std::min<u32>(nblocks, (u32)ctx->cus * 32)), dim3(64), 0, ctx->stream, b->desc, b->n, nblocks, cc, ctx->d_ticket, ctx->d_ticket + 8,
                           lists, sdesc, view);
    }
    HIPCHK(ctx, hipGetLastError());
    cs->masked = masked;
    cs->pretend = pretend;
    if (e0 && e1) {
        (void)hipEventRecord(e1, ctx->stream);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&cs->build_ms, e0, e1);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    // the bulk's view of the batch
    if (!cs->view) cs->view = new (std::nothrow) bsk_batch();
    if (!cs->view) return BSK_ERR_NOMEM;
    {
        bsk_batch *v = cs->view;
        u64 *bd = v->bdesc;  // its own binned copies survive (grow-only)
        u8 *bf = v->bflags;
        const size_t cbd = v->c_bdesc, cbf = v->c_bflags;
        *v = *b;
        v->hist = nullptr;
        v->borrowed = true;
        v->odd = nullptr;
        v->d_odd = nullptr;
        if (!masked) v->desc = view;  // (masked: the batch's own descriptors, the kernel masks by length)
        v->side_maxlen = b->maxlen;
        v->maxlen = bk.hi;
        v->n_bases = pretend ? (u64)pretend * b->n : bk.bases;
        v->uniform_len = pretend;
        v->bdesc = bd;
        v->bflags = bf;
        v->c_bdesc = cbd;
        v->c_bflags = cbf;
        v->bin_gran = 0;
        v->bin_lo = 0;
        v->bin_early = false;
        v->spare_ascii = nullptr;
        v->spare_aoff = nullptr;
    }
    cs->n = b->n;
    cs->n_bases = b->n_bases;
    cs->maxlen = b->maxlen;
    cs->desc = b->desc;
    cs->words = b->words;
    cs->blo = cuts[(size_t)bulk].lo;
    cs->bhi = cuts[(size_t)bulk].hi;
    return BSK_OK;
}

// what ran, for bsk_result_plan: the bulk's kernel + every part's
static void class_plan_names(bsk_result *res, const ClassSet *cs) {
    size_t at = strlen(res->plan);
    for (const auto &pt : cs->parts) {
        if (at + 8 >= sizeof res->plan) break;
        at += (size_t)snprintf(res->plan + at, sizeof res->plan - at, " + %s [%llu reads of %u..%u bases]", pt.res->plan, (unsigned long long)pt.n, pt.lo, pt.hi);
        at = std::min(at, sizeof res->plan - 1);
    }
}

// *applied = false: the batch keeps one plan (the caller goes on as before)
int run_classed(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p, int circ_ext, bsk_result **result, int warmup, int iters, float *kernel_ms, bool *applied) {
    *applied = false;
    const bool sizing = *result == nullptr || warmup + iters == 0;
    if (!sizing) {  // bsk_sketch_timed on a sized result: the class plan it was sized with, or none
        ClassSet *cs = (*result)->classes;
        if (!cs) return BSK_OK;
        if (ctx->cls_owner != *result || cs->n != b->n || cs->n_bases != b->n_bases || cs->maxlen != b->maxlen || cs->desc != b->desc || cs->words != b->words) {
            ctx->err = "bsk_sketch_timed: the result's class plan belongs to another batch (or a later bsk_sketch on this context replaced it): call bsk_sketch first";
            return BSK_ERR_ARG;
        }
        *applied = true;
        if (!side_ctx(ctx)) return fail_arg(ctx, "class plan: no side context");
        ctx->cls = cs;
        int rc = run_planned_resizing(ctx, cs->view, p, 0, result, warmup, iters, kernel_ms);
        ctx->cls = nullptr;
        if (rc == BSK_REPLAN_CLASS) {  // a part outgrew what its own sizing launch used: the whole plan is sized again, the parts with twice their regions, and the timed launches repeat (once)
            ctx->in_resize = ctx->part_grow = true;
            bool again = false;
            rc = run_classed(ctx, b, p, circ_ext, result, 0, 0, nullptr, &again);
            ctx->part_grow = false;
            if (rc == BSK_OK && !again) {
                ctx->err = "class plan: the batch no longer takes a class plan";
                rc = BSK_ERR_ARG;
            }
            if (rc == BSK_OK) {
                cs = (*result)->classes;
                ctx->cls = cs;
                rc = run_planned(ctx, cs->view, p, 0, result, warmup, iters, kernel_ms);
                ctx->cls = nullptr;
            }
            ctx->in_resize = false;
        }
        if (rc == BSK_OK) class_plan_names(*result, cs);
        return rc;
    }
    std::vector<ClassCut> cuts;
    int bulk = 0;
    if (!class_decide(ctx, b, p, circ_ext, cuts, bulk)) {
        if (*result && (*result)->classes) {
            class_set_free((*result)->classes);
            (*result)->classes = nullptr;
        }
        return BSK_OK;
    }
    ClassSet *cs = (*result && (*result)->classes) ? (*result)->classes : new (std::nothrow) ClassSet();
    if (!cs) return BSK_ERR_NOMEM;
    if (*result) (*result)->classes = nullptr;  // (held here until the run succeeded)
    ctx->cls_owner = nullptr;
    auto drop = [&](int code) {
        class_set_free(cs);
        return code;
    };
    bsk_ctx *const side = side_ctx(ctx);
    if (!side) return drop(fail_arg(ctx, "class plan: no side context"));
    int rc = class_build(ctx, b, p, cuts, bulk, cs);
    if (rc != BSK_OK) return drop(rc);
    {  // the lists and descriptors of the parts are in place: the side stream may read them
        const hipError_t se = hipStreamSynchronize(ctx->stream);
        if (se != hipSuccess) return drop(fail_hip(ctx, se, "class plan: hipStreamSynchronize"));
    }
    for (auto &pt : cs->parts) pt.sub->ctx = side;
    // every part sized as a batch of its own; then the parent, with the parts' slabs as its tail.  The parent's launch runs every part
    // AGAIN (into the tail): when one of them needs more room than its sizing launch did (BSK_REPLAN_CLASS), the parts are sized once more
    // with twice their overflow regions; after that the batch keeps one plan.
    for (int round = 0;; ++round) {
    ctx->cls_round = round;
    const bool grow = round > 0 || ctx->part_grow;
    u64 tail = 0;
    for (auto &pt : cs->parts) {
        const bool was_resize = side->in_resize;
        side->in_resize = grow && pt.res && !pt.tiled;  // (run_planned: twice the previous overflow region)
        struct Restore {
            bsk_ctx *c;
            bool v;
            ~Restore() { c->in_resize = v; }
        } restore{side, was_resize};
        if (pt.res && pt.res->arrays_borrowed) {  // a part of an earlier call: its place in that call's tail may be gone
            pt.res->hash = nullptr;
            pt.res->pos = nullptr;
            pt.res->arrays_borrowed = false;
            pt.res->cap = pt.res->alloc_cap = 0;
        }
        if (pt.tiled) {
            if (pt.res && !pt.res->wfirst) {  // (the part object of an earlier call that was not tile work)
                bsk_result_release(pt.res);
                pt.res = nullptr;
            }
            if (pt.res && pt.res->ctx != side) {
                bsk_result_release(pt.res);
                pt.res = nullptr;
            }
            side->tile_async = !grow;  // (sized again after an overflow: the round-trip path, which sizes by what the batch needs)
            side->tile_sync = grow;
            rc = sketch_tiled(side, pt.sub, p, 0, &pt.res, 0, 0, nullptr);
            pt.async = side->tile_was_async;
            side->tile_async = side->tile_sync = false;
            if (rc != BSK_OK) {
                ctx->err = side->err;
                return drop(rc);
            }
            pt.fresh = true;
            pt.extent = (pt.res->n_tuples + 31) & ~(u64)15;
            pt.off = tail;
            tail += pt.extent;
            continue;
        }
        if (pt.res && pt.res->wfirst) {  // (a wide result of an earlier call)
            bsk_result_release(pt.res);
            pt.res = nullptr;
        }
        if (pt.res && pt.res->ctx != side) {
            bsk_result_release(pt.res);
            pt.res = nullptr;
        }
        rc = run_planned(side, pt.sub, p, 0, &pt.res, 0, 0, nullptr);
        if (rc != BSK_OK) {
            ctx->err = side->err;
            return drop(rc);
        }
        pt.extent = (pt.res->cap + 15) & ~(u64)15;
        pt.off = tail;
        tail += pt.extent;
    }
    cs->tail = tail;
    ctx->cls = cs;
    rc = run_planned_resizing(ctx, cs->view, p, 0, result, warmup, iters, kernel_ms);
    ctx->cls = nullptr;
    if (rc == BSK_REPLAN_CLASS && round == 0) continue;
    if (rc == BSK_REPLAN_CLASS) {  // twice: this batch's parts do not hold still -- one plan for the whole batch (the caller's next step)
        class_set_free(cs);
        return BSK_OK;
    }
    break;
    }
    if (rc != BSK_OK) return drop(rc);
    bsk_result *res = *result;
    res->classes = cs;
    ctx->cls_owner = res;
    class_plan_names(res, cs);
    *applied = true;
    return BSK_OK;
}

