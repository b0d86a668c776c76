// kernels_host.hpp -- the host side's small utility kernels (synthetic batches, packing, flags, class cuts, digests, gathers, binned views).
// Internal linkage: every host translation unit (biosketch / planner / launch / tiles / classes .hip) that includes this header gets the ones
// it launches.
#pragma once
#include "kernels_generic.hpp"

using namespace bsk;

// ------------------------------------------------------------------------------------
// utility kernels
// ------------------------------------------------------------------------------------
static __global__ void k_synth_dna(u32 *words, u64 *desc, u8 *rflags, u64 n, u32 len, u32 wpr, u64 seed) {
    const u64 total = n * wpr;
    for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (u64)gridDim.x * blockDim.x) {
        const u64 r = g / wpr;
        const u32 j = (u32)(g - r * wpr);
        u32 v = (u32)splitmix64(seed + g);
        const u32 valid = len - j * 16u;
        if (valid < 16u) v &= (1u << (2 * valid)) - 1u;
        words[g] = v;
        if (j == 0) {
            desc[r] = ((r * wpr) << 24) | len;
            rflags[r] = 0;
        }
    }
}
static __global__ void k_synth_protein(u8 *ascii, u64 *aoff, u64 n, u32 len, u64 seed) {
    const u64 total = n * len;
    const char *aa = "ACDEFGHIKLMNPQRSTVWY";
    for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (u64)gridDim.x * blockDim.x) {
        ascii[g] = (u8)aa[splitmix64(seed + g) % 20u];
        if (g <= n) aoff[g] = g * len;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && total < n + 1) {
        for (u64 g = total; g <= n; ++g) aoff[g] = g * len;
    }
}
// ASCII -> 2-bit words.  One thread per output word; the owning read is found by a
// binary search over desc[] (first_word is monotone).  Not on the hot path.
static __global__ void k_pack(const u8 *ascii, const u64 *aoff, const u64 *desc, const u64 *fw, u64 n, u64 n_words, u32 *words, u8 *rflags,
                       u32 *nonacgt_reads, u32 *wbits) {
    for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g < n_words; g += (u64)gridDim.x * blockDim.x) {
        u64 lo = 0, hi = n - 1;  // largest r with first_word[r] <= g  (reads with 0 words share a first_word: take the last)
        while (lo < hi) {
            u64 mid = (lo + hi + 1) >> 1;
            if ((desc ? desc[mid] >> 24 : fw[mid]) <= g) lo = mid;
            else hi = mid - 1;
        }
        const u64 L = aoff[lo + 1] - aoff[lo];
        const u64 j = g - (desc ? desc[lo] >> 24 : fw[lo]);
        const u8 *src = ascii + aoff[lo] + j * 16;
        const u64 nb = L > j * 16 ? (L - j * 16 < 16 ? L - j * 16 : 16) : 0;
        u32 v = 0;
        bool bad = false;
        for (u64 b = 0; b < nb; ++b) {
            unsigned c = acgt_code(src[b]);
            if (c > 3) {
                bad = true;
                c = 0;
            }
            v |= c << (2 * b);
        }
        words[g] = v;
        if (bad) {
            if (wbits) atomicOr(&wbits[g >> 5], 1u << (g & 31));
            if (rflags[lo] == 0) atomicAdd(nonacgt_reads, 1u);  // approximate under races; recounted on host
            rflags[lo] = BSK_ST_HAS_NON_ACGT;
        }
    }
}
static __global__ void k_count_flags(const u8 *rflags, u64 n, u32 *count) {
    u32 c = 0;
    for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (u64)gridDim.x * blockDim.x)
        c += rflags[g] != 0;
    for (int d = 32; d; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}
// indices of the flagged reads, ascending (one wavefront per 64 reads + look-back: deterministic order)
static __global__ __launch_bounds__(64) void k_compact_flags(const u8 *rflags, u64 n, u32 nunits, u32 *ticket, u64 *lookback, u32 *subset) {
    const int lane = lane_id();
    HeadTickets tickets(reinterpret_cast<u32 *>(lookback + lb_heads_at(nunits)));  // (a unit is 64 flag bytes: one head word would BE the kernel -- 10^8 reads = 1.6 10^6 tickets = 18 ms)
    for (;;) {
        const u32 unit = tickets.next(nunits, lane);
        if (unit == ~0u) break;
        const u64 r = (u64)unit * 64 + lane;
        const bool f = r < n && rflags[r] != 0;
        const u64 m = __ballot(f);
        const u64 base = lookback_exclusive(lookback, unit, (u64)__builtin_popcountll(m), lane);
        if (f) subset[base + __builtin_popcountll(m & ((1ULL << lane) - 1))] = (u32)r;
    }
}
// Class plans (run_classed): ONE pass over the descriptors cuts the batch -- every read of a class other than the bulk is appended to its
// class's list with its descriptor next to it (the class then runs as a batch of its own over the parent's words), and the BULK's view of
// the batch is written: a read of another class keeps its place and its first word and PRETENDS the bulk's length -- `pretend` bases when
// every read of the bulk has that length (the view stays a fixed-length batch: the kernels' fast paths; what the bulk's kernel makes of
// such a read's first bases is overwritten by the part that owns it; reading past a shorter read stays inside words[]: pad_words), 0 bases
// otherwise (an empty SHORT entry).  A ticket is 16 rows of 64 reads; a class's place in its list comes from ONE atomic per ticket and
// class present (a decoupled look-back per class and ticket was latency-bound: 0.6-0.9 ms per class for 4 10^7 reads).  The lists are in
// arrival order: which slab of a part a read gets may differ from run to run, what it holds does not.
struct ClassCuts {
    u32 hi[8];     // class c takes the lengths (hi[c-1], hi[c]]
    u32 first[8];  // where class c's list starts in list[] / sdesc[] (exact counts are known on the host: LenHist)
    u32 ncls, bulk, pretend, pad;
};
static __device__ __forceinline__ u32 class_of(const ClassCuts &cc, u32 len) {
    u32 c = 0;
#pragma unroll
    for (int q = 0; q < 7; ++q) c += (q + 1 < (int)cc.ncls && len > cc.hi[q]) ? 1u : 0u;
    return c;
}
static __global__ __launch_bounds__(64) void k_class_cut(const u64 *desc, u64 n, u32 nblocks, ClassCuts cc, u32 *ticket, u32 *cursor, u32 *list, u64 *sdesc, u64 *view) {
    constexpr int ROWS = 16;
    const int lane = lane_id();
    (void)ticket;
    for (u32 blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {  // (lists are in arrival order anyway: no ticket counter to queue at)
        const u64 r0 = (u64)blk * ROWS * 64 + lane;
        u32 c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;  // reads per class in this ticket (wave-uniform)
        bool any = false;
        u64 dd[ROWS];  // all sixteen rows are requested before the first is used (gfx9 counts loads and stores in ONE in-order vmcnt: a load
                       // issued behind the previous row's store to view[] waited for that store -- 16 round trips per ticket, 1.4 ms for 6.7 10^7 reads)
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const u64 r = r0 + (u64)j * 64;
            dd[j] = r < n ? desc[r] : 0;
        }
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {  // (the view's rows leave back to back: a store between two uses of loaded rows made the compiler wait for it)
            const u64 r = r0 + (u64)j * 64;
            const u64 d = dd[j];
            const u32 c = r < n ? class_of(cc, (u32)(d & 0xffffffULL)) : cc.bulk;
            view[r] = c == cc.bulk ? d : ((d & ~0xffffffULL) | cc.pretend);  // (unconditional: view[] has a ticket's worth of slack behind the batch -- a branch here costs a wait per row)
        }
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const u64 r = r0 + (u64)j * 64;
            const u64 d = dd[j];
            const u32 c = r < n ? class_of(cc, (u32)(d & 0xffffffULL)) : cc.bulk;
            if (__ballot(c != cc.bulk)) {  // (rare for outlier classes: most rows are all bulk)
                any = true;
                c0 += (u32)__builtin_popcountll(__ballot(c == 0u));
                c1 += (u32)__builtin_popcountll(__ballot(c == 1u));
                c2 += (u32)__builtin_popcountll(__ballot(c == 2u));
                c3 += (u32)__builtin_popcountll(__ballot(c == 3u));
                c4 += (u32)__builtin_popcountll(__ballot(c == 4u));
                c5 += (u32)__builtin_popcountll(__ballot(c == 5u));
                c6 += (u32)__builtin_popcountll(__ballot(c == 6u));
                c7 += (u32)__builtin_popcountll(__ballot(c == 7u));
            }
        }
        if (!any) continue;
        // this ticket's place in every list it adds to: one atomic per class present (lane q asks for class q)
        u32 mine = lane == 0 ? c0 : lane == 1 ? c1 : lane == 2 ? c2 : lane == 3 ? c3 : lane == 4 ? c4 : lane == 5 ? c5 : lane == 6 ? c6 : lane == 7 ? c7 : 0u;
        if (lane >= 8 || (u32)lane == cc.bulk) mine = 0;
        u32 at = 0;
        if (mine) at = cc.first[lane & 7] + atomicAdd(&cursor[lane & 7], mine);
        for (int j = 0; j < ROWS; ++j) {  // (the rows again, from the L2: nothing is kept across the two passes)
            const u64 r = r0 + (u64)j * 64;
            const u64 d = r < n ? desc[r] : 0;
            const u32 c = r < n ? class_of(cc, (u32)(d & 0xffffffULL)) : cc.bulk;
            u64 others = __ballot(c != cc.bulk);
            while (others) {  // every class present in the row, lowest first
                const int src = __builtin_ctzll(others);
                const u32 q = (u32)__builtin_amdgcn_readlane((int)c, src);
                const u64 m = __ballot(c == q);
                const u32 base = (u32)__builtin_amdgcn_readlane((int)at, (int)q);
                if (c == q) {
                    const u32 i = base + (u32)__builtin_popcountll(m & ((1ULL << lane) - 1));
                    list[i] = (u32)r;
                    sdesc[i] = d;
                }
                if ((u32)lane == q) at += (u32)__builtin_popcountll(m);
                others &= ~m;
            }
        }
    }
}
// descriptors of a host-built class list
static __global__ void k_gather_desc(const u64 *desc, const u32 *list, u64 n, u64 *sdesc) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) sdesc[i] = desc[list[i]];
}
// The lists of a class plan from the batch's list of odd sequences on the device (bsk_batch::d_odd: index << 32 | length, ascending): a
// wavefront takes 64 entries and claims room in every class's list with one atomic per class present (cursor[c]); the descriptors are
// gathered on the way.  (The same on the host -- a loop over the list and one copy -- is 0.9 ms for the 10^6 odd reads of a batch of
// 10^8 with 1 % of 250-base reads, 7 % of its kernel, on every bsk_sketch.)  Order inside a class: ascending inside a wavefront's 64.
static __global__ __launch_bounds__(256) void k_odd_split(const u64 *odd, u64 n_odd, ClassCuts cc, u32 n_out, u32 *cursor, const u64 *desc, u32 *lists, u64 *sdesc) {
    const u32 lane = threadIdx.x & 63u;
    for (u64 i0 = ((u64)blockIdx.x * 256 + (threadIdx.x & ~63u)); i0 < n_odd; i0 += (u64)gridDim.x * 256) {
        const u64 i = i0 + lane;
        const u64 e = i < n_odd ? odd[i] : 0;
        const u32 c = i < n_odd ? class_of(cc, (u32)e) : cc.bulk;
        for (u32 q = 0; q < cc.ncls; ++q) {
            if (q == cc.bulk) continue;
            const u64 m = __builtin_amdgcn_ballot_w64(c == q);
            if (!m) continue;
            u32 base = 0;
            if (lane == (u32)__builtin_ctzll(m)) base = atomicAdd(&cursor[q], (u32)__builtin_popcountll(m));
            base = (u32)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(m));
            if (c == q) {
                const u32 at = cc.first[q] + base + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0));
                const u32 end = q + 1 < cc.ncls ? cc.first[q + 1] : n_out;  // (first[] of the bulk's successor skips nothing: the bulk has no list)
                if (at < end) {
                    lists[at] = (u32)(e >> 32);
                    sdesc[at] = desc[e >> 32];
                }
            }
        }
    }
}
// the reads of a part take their reference words (re-based into the parent's tail) and status bytes from the part's result
static __global__ void k_adopt_refs(const u32 *list, u64 n, const u64 *crefs, const u8 *cstatus, u64 base, u64 *refs, u8 *status) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 r = list[i], c = crefs[i];
        const u64 first = (c & ~BSK_REF_ROWS) >> 24;
        refs[r] = (c & BSK_REF_ROWS) | ((first + base) << 24) | (c & 0xffffffULL);
        status[r] = cstatus[i];
    }
}
// the ASCII side launch ran beside the main kernel with reference words and status bytes of its own (indexed like the batch): they
// replace the main kernel's for the reads of the subset
// A part of a class plan has run on the side context: its overflow flags (ticket words 1 and 3 of THAT context, reset by the next part's
// launch) are folded into one word of the parent's, which the parent's read-backs look at (a part that overflowed on the
// launch the caller sees -- region and list use vary from launch to launch -- was adopted with truncated tuples and no error).
static __global__ void k_fold_flags(const u32 *side_ticket, u32 *parent_word) {
    const u32 f = side_ticket[1] | side_ticket[3];
    if (f) atomicOr(parent_word, f);
}
// sketch_tiled without a last synchronisation (a class plan's tiled part): the tile kernels' overflow flags (saved words 1 and 3) and the
// stitch's (word 1 of the live ticket) become one word the parent folds into its own (launch_parts)
static __global__ void k_tile_flag_word(const u32 *saved, const u32 *live, u32 *out) { out[0] = saved[1] | saved[3] | live[1]; }
static __global__ void k_fold_word(const u32 *word, u32 *parent_word) {
    if (word[0]) atomicOr(parent_word, word[0]);
}
static __global__ void k_adopt_side(const u32 *subset, u64 nsub, const u64 *srefs, const u8 *sstatus, u64 *refs, u8 *status) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < nsub; i += (u64)gridDim.x * blockDim.x) {
        const u64 r = subset[i];
        refs[r] = srefs[r];
        status[r] = sstatus[r];
    }
}
// ... the same from a part that ran over tiles (a wide result: first / count per sequence)
static __global__ void k_adopt_wide(const u32 *list, u64 n, const u64 *wfirst, const u64 *wcount, const u8 *cstatus, u64 base, u64 *refs, u8 *status) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 r = list[i];
        refs[r] = ((wfirst[i] + base) << 24) | (wcount[i] & 0xffffffULL);
        status[r] = cstatus[i];
    }
}
// circular: read r' = read r + its first k-1 bases (iterator.go:642-646).  One thread per output word.
// Source / destination sequences are located by packed descriptors (desc: (first_word << 24) | bases) or, when a sequence has
// 2^24 bases or more, by first-word + length arrays (fw / llen).
static __global__ void k_extend_packed(const u32 *words, const u64 *desc, const u64 *fw, const u64 *llen, const u64 *ndesc, const u64 *nfw,
                                const u64 *nllen, u64 n, u64 n_words_new, u32 *out) {
    for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g < n_words_new; g += (u64)gridDim.x * blockDim.x) {
        u64 lo = 0, hi = n - 1;
        while (lo < hi) {
            u64 mid = (lo + hi + 1) >> 1;
            if ((ndesc ? (ndesc[mid] >> 24) : nfw[mid]) <= g) lo = mid;
            else hi = mid - 1;
        }
        const u64 L = desc ? (desc[lo] & 0xffffffULL) : llen[lo], L2 = ndesc ? (ndesc[lo] & 0xffffffULL) : nllen[lo];
        const u32 *src = words + (desc ? (desc[lo] >> 24) : fw[lo]);
        const u64 j0 = (g - (ndesc ? (ndesc[lo] >> 24) : nfw[lo])) * 16;
        u32 v = 0;
        for (u64 b = 0; b < 16 && j0 + b < L2; ++b) {
            u64 p = j0 + b;
            if (p >= L) p -= L;
            v |= ((src[p >> 4] >> ((p & 15) * 2)) & 3u) << (2 * b);
        }
        out[g] = v;
    }
}
static __global__ void k_extend_ascii(const u8 *ascii, const u64 *aoff, const u64 *naoff, u64 n, u8 *out) {
    // one wave per read
    const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 r = wave; r < n; r += nw) {
        const u64 L = aoff[r + 1] - aoff[r], L2 = naoff[r + 1] - naoff[r];
        for (u64 p = threadIdx.x & 63; p < L2; p += 64) out[naoff[r] + p] = ascii[aoff[r] + (p < L ? p : p - L)];
    }
}

// digest: checksum = sum over tuples of hash*(2*position+1) -- a sum, so any traversal will do.  A wavefront takes 64 reads: when they
// are stored as unit rows a lane walks its own read (row t of the unit is one coalesced load); otherwise (slabs, per-read runs: a
// lane's tuples are contiguous and the lanes' runs 256 bytes or more apart -- 64 lines per load, 50 ms for configs[2]'s result) the
// reads are taken four at a time by 16 lanes each, whole 128-byte pieces of a run per load.
static __global__ void k_digest(const u64 *hash, const u32 *pos, const u64 *refs, const u64 *wfirst, const u64 *wcount, u64 n,
                         u64 *out /*[0] checksum [1] tuples*/) {
    u64 s = 0, c = 0;
    const int lane = threadIdx.x & 63;
    const u64 nblk = (n + 63) / 64;
    for (u64 blk = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6; blk < nblk; blk += ((u64)gridDim.x * blockDim.x) >> 6) {
        const u64 r = blk * 64 + lane;
        u64 b = 0, cnt = 0, st = 1;
        if (r < n) {
            b = refs ? BSK_REF_FIRST(refs[r]) : wfirst[r];
            cnt = refs ? BSK_REF_COUNT(refs[r]) : wcount[r];
            st = refs ? BSK_REF_STRIDE(refs[r]) : 1;
        }
        c += cnt;
        if (st != 1)  // unit rows (a listed read of such a unit lies elsewhere with stride 1 and is taken below)
            for (u64 t = 0; t < cnt; ++t) s += hash[b + t * st] * (2ULL * (pos ? (u64)(pos[b + t * st] & BSK_POS_MASK) : t) + 1ULL);
        if (__builtin_amdgcn_ballot_w64(cnt > 0 && st == 1) == 0) continue;
        if (st != 1) cnt = 0;
        for (int i = 0; i < 16; ++i) {
            const int a = (i * 4 + (lane >> 4)) << 2;
            const u64 bq = ((u64)(u32)__builtin_amdgcn_ds_bpermute(a, (int)(u32)(b >> 32)) << 32) | (u32)__builtin_amdgcn_ds_bpermute(a, (int)(u32)b);
            const u64 cq = ((u64)(u32)__builtin_amdgcn_ds_bpermute(a, (int)(u32)(cnt >> 32)) << 32) | (u32)__builtin_amdgcn_ds_bpermute(a, (int)(u32)cnt);
            const u64 sq = (u32)__builtin_amdgcn_ds_bpermute(a, (int)(u32)st);
            for (u64 t = (u64)(lane & 15); t < cq; t += 16) s += hash[bq + t * sq] * (2ULL * (pos ? (u64)(pos[bq + t * sq] & BSK_POS_MASK) : t) + 1ULL);
        }
    }
    s = wave_sum_u64(s);
    c = wave_sum_u64(c);
    if (lane == 0) {
        atomicAdd(&out[0], s);
        atomicAdd(&out[1], c);
    }
}
static __global__ void k_sum_counts(const u64 *refs, u64 n, u64 *out) {
    u64 c = 0;
    for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (u64)gridDim.x * blockDim.x) c += refs[r] & 0xffffffULL;
    c = wave_sum_u64(c);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}
// fetch: pack the tuples of reads [first, first+count) densely (dst offsets computed on the host); one wave per read
static __global__ void k_gather(const u64 *hash, const u32 *pos, const u64 *refs, const u64 *wfirst, const u64 *wcount, const u64 *dstoff,
                         u64 count, u64 *ohash, u32 *opos) {
    const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 r = wave; r < count; r += nw) {
        const u64 b = refs ? BSK_REF_FIRST(refs[r]) : wfirst[r], cnt = refs ? BSK_REF_COUNT(refs[r]) : wcount[r], d = dstoff[r];
        const u64 st = refs ? BSK_REF_STRIDE(refs[r]) : 1;
        for (u64 t = threadIdx.x & 63; t < cnt; t += 64) {
            if (ohash) ohash[d + t] = hash[b + t * st];
            if (opos) opos[d + t] = pos[b + t * st];
        }
    }
}
static __global__ void k_digest_status(const u8 *status, u64 n, u64 *out4) {
    u64 c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (u64)gridDim.x * blockDim.x) {
        const u8 s = status[g];
        c0 += (s & BSK_ST_CODE_MASK) == BSK_ST_SHORT;
        c1 += (s & BSK_ST_CODE_MASK) == BSK_ST_ILLEGAL;
        c2 += (s & BSK_ST_FIRST_WINDOW_TIE) != 0;
        c3 += (s & BSK_ST_HAS_NON_ACGT) != 0;
    }
    c0 = wave_sum_u64(c0);
    c1 = wave_sum_u64(c1);
    c2 = wave_sum_u64(c2);
    c3 = wave_sum_u64(c3);
    if ((threadIdx.x & 63) == 0) {
        if (c0) atomicAdd(&out4[0], c0);
        if (c1) atomicAdd(&out4[1], c1);
        if (c2) atomicAdd(&out4[2], c2);
        if (c3) atomicAdd(&out4[3], c3);
    }
}

// Length binning (KArgs::binned).  The packed minimizer / syncmer kernels walk the 64 reads of a unit in lock-step, so a unit costs its
// LONGEST read: trimmed reads (lengths 60..150) ran at 0.64-0.73 of the fixed-length rate.  Here the reads of every chunk of 4096 -- 64
// units -- are stably ordered by length class, so that the reads of a unit end together.  The class is the number of `gran`-wide steps
// the kernel takes over the read: ceil((bases - lo) / gran) (a plan's own view: lo = k - 1, gran = a multiple of the kernel's block of w k-mers,
// at most 64 classes; the view built with the batch, bin_with_batch: lo = the shortest read - 1, gran = 1 base where the lengths span 126 or
// fewer, 128 classes; slots beyond the batch sort last).  bdesc[4096 c + j] = the descriptor of chunk c's j-th read in that order | the read's
// own place in the chunk << 12 (batches of reads shorter than 4096 bases: bits 12..23 of a descriptor are free); bflags follows rflags.
// The kernels write the reference word and status byte of a read at its own place (out_index, kernels_generic.hpp): the permutation
// never leaves a chunk, i.e. 32 KB of reference words written by a few wavefronts at about the same time.
// One workgroup of 512 per chunk: wave v takes rows 8 v .. 8 v + 7 (a row = 64 consecutive reads); stable ranks inside a row come from
// ballots, class by class; cnt[row][class] is scanned over the rows by 65 threads and the class totals by one wavefront.
// (mlo / mhi / mpretend: a class plan's bulk over the whole batch -- desc_len(), kernels_generic.hpp: the sequences of the other classes enter
// with the pretended length, so that the bits of a length of 4096 or more never reach the place field)
template <int NC>  // classes: 64 (a plan's own view), 128 (the view built with the batch)
static __global__ __launch_bounds__(512) void k_bin_desc(const u64 *desc, const u8 *rflags, u64 n, u32 lo, u32 gran, u64 *bdesc, u8 *bflags, u32 mlo, u32 mhi,
                                                  u32 mpretend) {
    __shared__ u32 cnt[64][NC + 2];  // reads of the class in the row, then the first place of that run inside its class
    __shared__ u32 tot[NC + 2];      // reads of the class in the chunk, then the class's first place in the chunk
    __shared__ u32 wsum[NC / 64];
    const u32 tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const u64 nchunks = (n + 4095) / 4096;
    for (u64 c = blockIdx.x; c < nchunks; c += gridDim.x) {
        for (u32 i = tid; i < 64 * (NC + 2); i += 512) (&cnt[0][0])[i] = 0;
        __syncthreads();
        u64 d[8];
        u32 cls[8], rank[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const u32 row = wave * 8 + j;
            const u64 i = c * 4096 + row * 64 + lane;
            d[j] = i < n ? desc[i] : 0;
            u32 cl = NC;
            if (i < n) {
                u32 L = (u32)(d[j] & 0xffffffULL);
                if (mhi && (L < mlo || L > mhi)) {
                    L = mpretend;
                    d[j] = (d[j] & ~0xffffffULL) | L;
                }
                cl = L > lo ? (L - lo + gran - 1) / gran : 0u;
                cl = cl < (u32)(NC - 1) ? cl : (u32)(NC - 1);
            }
            u32 rk = 0;
            for (u64 todo = ~0ULL; todo;) {
                const int first = __builtin_amdgcn_readfirstlane(__builtin_ctzll(todo));
                const u32 cc = (u32)__builtin_amdgcn_readlane((int)cl, first);
                const u64 m = __builtin_amdgcn_ballot_w64(cl == cc);
                if (cl == cc) rk = __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0));
                if ((int)lane == first) cnt[row][cc] = (u32)__builtin_popcountll(m);
                todo &= ~m;
            }
            cls[j] = cl;
            rank[j] = rk;
        }
        __syncthreads();
        if (tid < NC + 1) {
            u32 run = 0;
#pragma unroll 8
            for (int r = 0; r < 64; ++r) {
                const u32 t = cnt[r][tid];
                cnt[r][tid] = run;
                run += t;
            }
            tot[tid] = run;
        }
        __syncthreads();
        {  // (class NC = the slots beyond the batch: behind everything else)
            u32 t = 0, inc = 0;
            if (tid < NC) {
                t = tot[tid];
                inc = wave_incl_scan_u32(t, (int)lane);
                if (lane == 63) wsum[wave] = inc;
            }
            __syncthreads();
            if (tid < NC) {
                u32 before = 0;
                for (u32 q = 0; q < wave; ++q) before += wsum[q];
                tot[tid] = before + inc - t;
                if (tid == NC - 1) tot[NC] = before + inc;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const u32 row = wave * 8 + j;
            const u64 i = c * 4096 + row * 64 + lane;
            if (i < n) {
                const u64 dest = c * 4096 + tot[cls[j]] + cnt[row][cls[j]] + rank[j];
                bdesc[dest] = d[j] | ((u64)(row * 64 + lane) << 12);
                if (rflags) bflags[dest] = rflags[i];
            }
        }
        __syncthreads();
    }
}

