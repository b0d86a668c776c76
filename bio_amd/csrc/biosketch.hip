// biosketch.hip -- C ABI (include/biosketch.h) of the MI355X k-mer sketching engine.
// Host side: contexts, batches (device-resident 2-bit packed reads), results
// (device-resident CSR tuples) and the kernel dispatch.  gfx950 only; there is no
// CPU implementation behind this ABI: without a device every compute entry fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "biosketch.h"
#include "host_types.hpp"
#include "kernels_generic.hpp"
#include "kernels_fast.hpp"
#include "kernels_more.hpp"
#include "fast_dispatch.hpp"
#include "planner_table.hpp"
#include "kernels_translate.hpp"
#include "kernels_tile.hpp"
#include "kernels_simhash.hpp"

using namespace bsk;

// ------------------------------------------------------------------------------------
// utility kernels
// ------------------------------------------------------------------------------------
__global__ void k_synth_dna(u32 *words, u64 *desc, u8 *rflags, u64 n, u32 len, u32 wpr, u64 seed) {
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
__global__ void k_synth_protein(u8 *ascii, u64 *aoff, u64 n, u32 len, u64 seed) {
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
__global__ void k_pack(const u8 *ascii, const u64 *aoff, const u64 *desc, const u64 *fw, u64 n, u64 n_words, u32 *words, u8 *rflags,
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
__global__ void k_count_flags(const u8 *rflags, u64 n, u32 *count) {
    u32 c = 0;
    for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (u64)gridDim.x * blockDim.x)
        c += rflags[g] != 0;
    for (int d = 32; d; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}
// indices of the flagged reads, ascending (one wavefront per 64 reads + look-back: deterministic order)
__global__ __launch_bounds__(64) void k_compact_flags(const u8 *rflags, u64 n, u32 nunits, u32 *ticket, u64 *lookback, u32 *subset) {
    const int lane = lane_id();
    for (;;) {
        const u32 unit = next_ticket(ticket, lane);
        if (unit >= nunits) break;
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
__device__ __forceinline__ u32 class_of(const ClassCuts &cc, u32 len) {
    u32 c = 0;
#pragma unroll
    for (int q = 0; q < 7; ++q) c += (q + 1 < (int)cc.ncls && len > cc.hi[q]) ? 1u : 0u;
    return c;
}
__global__ __launch_bounds__(64) void k_class_cut(const u64 *desc, u64 n, u32 nblocks, ClassCuts cc, u32 *ticket, u32 *cursor, u32 *list, u64 *sdesc, u64 *view) {
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
__global__ void k_gather_desc(const u64 *desc, const u32 *list, u64 n, u64 *sdesc) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) sdesc[i] = desc[list[i]];
}
// The lists of a class plan from the batch's list of odd sequences on the device (bsk_batch::d_odd: index << 32 | length, ascending): a
// wavefront takes 64 entries and claims room in every class's list with one atomic per class present (cursor[c]); the descriptors are
// gathered on the way.  (The same on the host -- a loop over the list and one copy -- is 0.9 ms for the 10^6 odd reads of a batch of
// 10^8 with 1 % of 250-base reads, 7 % of its kernel, on every bsk_sketch.)  Order inside a class: ascending inside a wavefront's 64.
__global__ __launch_bounds__(256) void k_odd_split(const u64 *odd, u64 n_odd, ClassCuts cc, u32 n_out, u32 *cursor, const u64 *desc, u32 *lists, u64 *sdesc) {
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
__global__ void k_adopt_refs(const u32 *list, u64 n, const u64 *crefs, const u8 *cstatus, u64 base, u64 *refs, u8 *status) {
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
__global__ void k_fold_flags(const u32 *side_ticket, u32 *parent_word) {
    const u32 f = side_ticket[1] | side_ticket[3];
    if (f) atomicOr(parent_word, f);
}
// sketch_tiled without a last synchronisation (a class plan's tiled part): the tile kernels' overflow flags (saved words 1 and 3) and the
// stitch's (word 1 of the live ticket) become one word the parent folds into its own (launch_parts)
__global__ void k_tile_flag_word(const u32 *saved, const u32 *live, u32 *out) { out[0] = saved[1] | saved[3] | live[1]; }
__global__ void k_fold_word(const u32 *word, u32 *parent_word) {
    if (word[0]) atomicOr(parent_word, word[0]);
}
__global__ void k_adopt_side(const u32 *subset, u64 nsub, const u64 *srefs, const u8 *sstatus, u64 *refs, u8 *status) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < nsub; i += (u64)gridDim.x * blockDim.x) {
        const u64 r = subset[i];
        refs[r] = srefs[r];
        status[r] = sstatus[r];
    }
}
// ... the same from a part that ran over tiles (a wide result: first / count per sequence)
__global__ void k_adopt_wide(const u32 *list, u64 n, const u64 *wfirst, const u64 *wcount, const u8 *cstatus, u64 base, u64 *refs, u8 *status) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 r = list[i];
        refs[r] = ((wfirst[i] + base) << 24) | (wcount[i] & 0xffffffULL);
        status[r] = cstatus[i];
    }
}
// circular: read r' = read r + its first k-1 bases (iterator.go:642-646).  One thread per output word.
// Source / destination sequences are located by packed descriptors (desc: (first_word << 24) | bases) or, when a sequence has
// 2^24 bases or more, by first-word + length arrays (fw / llen).
__global__ void k_extend_packed(const u32 *words, const u64 *desc, const u64 *fw, const u64 *llen, const u64 *ndesc, const u64 *nfw,
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
__global__ void k_extend_ascii(const u8 *ascii, const u64 *aoff, const u64 *naoff, u64 n, u8 *out) {
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
__global__ void k_digest(const u64 *hash, const u32 *pos, const u64 *refs, const u64 *wfirst, const u64 *wcount, u64 n,
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
__global__ void k_sum_counts(const u64 *refs, u64 n, u64 *out) {
    u64 c = 0;
    for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (u64)gridDim.x * blockDim.x) c += refs[r] & 0xffffffULL;
    c = wave_sum_u64(c);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}
// fetch: pack the tuples of reads [first, first+count) densely (dst offsets computed on the host); one wave per read
__global__ void k_gather(const u64 *hash, const u32 *pos, const u64 *refs, const u64 *wfirst, const u64 *wcount, const u64 *dstoff,
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
__global__ void k_digest_status(const u8 *status, u64 n, u64 *out4) {
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
__global__ __launch_bounds__(512) void k_bin_desc(const u64 *desc, const u8 *rflags, u64 n, u32 lo, u32 gran, u64 *bdesc, u8 *bflags, u32 mlo, u32 mhi,
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

// ------------------------------------------------------------------------------------
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
    *out = ctx;
    return BSK_OK;
}

extern "C" void bsk_ctx_destroy(bsk_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
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
static int grid_for(bsk_ctx *ctx, u64 items, int block) {
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
static int ensure_binned(bsk_ctx *ctx, const bsk_batch *b, u32 lo, u32 gran, u32 mlo, u32 mhi, u32 mpretend, bool fine = false);
// The length-binned view of a ragged batch of short reads, built WITH the batch (its lengths are known there, the pass runs behind the
// pack kernel on the same stream): classes so fine -- (longest - shortest) / 63 bases, one or two bases for trimmed reads -- that the
// reads of a unit end within a base or two of each other whatever the plan's block of w k-mers or k - s s-mers is, so that no plan
// needs a pass of its own (round 4: k_bin_desc per plan was 10 % of the minimizer kernel on a fresh batch; bsk_batch_prepare now
// reports 0 for such a batch).  Only batches the planner can bin at all (bin_gran_for); BSK_NO_BIN_EARLY=1: per plan, as before.
static int bin_with_batch(bsk_ctx *ctx, bsk_batch *b) {
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
static constexpr u64 kMaxPrefetchWords = 32;  // >= SynPkLdsL::NW (static_assert beside pk_syncmer_max_bases, kernels_syncmer_pk.hpp)
// The smallest read length any kind tiles from (sketch_impl's tile_min: syncmers on the long packed plan from 448 bases, stream kinds from
// 16 (BSK_NT_FAST_WORDS - 2) = 512): batch creation keeps the non-ACGT word bits of every batch that MAY be tiled.
static constexpr u32 kSynTileMin = (u32)PlannerTable::syn_tile_min_bases;
static u32 min_tile_min(const bsk_ctx *ctx);
static u64 pad_words(u32 maxlen) { return (u64)maxlen / 16 + kMaxPrefetchWords + 1; }
static u32 env_u32(const char *name, u32 dflt) {
    const char *v = getenv(name);
    return v && *v ? (u32)strtoul(v, nullptr, 10) : dflt;
}
static u32 min_tile_min(const bsk_ctx *ctx) { return ctx->opt.tile_min ? ctx->opt.tile_min : std::min<u32>(64u, 16u * (BSK_NT_FAST_WORDS - 2)); }  // (syncmers and wide-window minimizers tile from where their staged kernel stops fitting: tile_min_for)
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
static bsk_ctx *side_ctx(bsk_ctx *ctx) {
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
static int build_subset(bsk_ctx *ctx, bsk_batch *b);

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
static void class_set_free(ClassSet *cs);
extern "C" void bsk_result_release(bsk_result *r) {
    if (!r) return;
    if (r->ctx) (void)hipSetDevice(r->ctx->device);
    if (r->classes) class_set_free(r->classes);
    if (r->ctx && r->ctx->cls_owner == r) r->ctx->cls_owner = nullptr;
    (void)hipFree(r->refs);
    (void)hipFree(r->wfirst);
    (void)hipFree(r->wcount);
    (void)hipFree(r->status);
    if (!r->arrays_borrowed) {
        (void)hipFree(r->hash);
        (void)hipFree(r->pos);
    }
    delete r;
}

static bool kind_has_pos(int kind) { return kind == BSK_MINIMIZER || kind == BSK_SYNCMER || kind == BSK_PROT_MINIMIZER; }

// tail: tuples reserved BEHIND the logical capacity `cap` (class plans: the slabs of the adopted parts live there; the kernels of the
// result itself never see them)
static int result_prepare(bsk_ctx *ctx, bsk_result **res, u64 n, int kind, u64 cap, u64 tail = 0) {
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
        if ((e = hipMalloc(&r->refs, (n ? n : 1) * 8)) != hipSuccess || (e = hipMalloc(&r->status, n ? n : 1)) != hipSuccess) {
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
            (void)hipFree(r->hash);
            (void)hipFree(r->pos);
        }
        r->arrays_borrowed = false;
        r->hash = nullptr;
        r->pos = nullptr;
        r->cap = 0;
        r->alloc_cap = 0;
        hipError_t e;
        if ((e = hipMalloc(&r->hash, (cap + tail + 2) * 8)) != hipSuccess) return fail_hip(ctx, e, "result hash alloc");
        if (hp && (e = hipMalloc(&r->pos, (cap + tail + 2) * 4)) != hipSuccess) return fail_hip(ctx, e, "result pos alloc");
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
// dispatch
// ------------------------------------------------------------------------------------
static int validate(const bsk_params *p, int alphabet) {
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

static int ensure_scratch(bsk_ctx *ctx, size_t nunits, size_t ring_entries) {
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
static int build_subset(bsk_ctx *ctx, bsk_batch *b) {
    (void)hipFree(b->subset);
    b->subset = nullptr;
    b->nsub = 0;
    if (!b->n || !b->n_nonacgt || !b->rflags || b->n >= (1ULL << 32)) return BSK_OK;
    const u32 nunits = (u32)((b->n + 63) / 64);
    int rc = ensure_scratch(ctx, nunits, 0);
    if (rc != BSK_OK) return rc;
    HIPCHK(ctx, hipMalloc(&b->subset, b->n_nonacgt * sizeof(u32)));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_ticket, 0, 4 * sizeof(u32), ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_lookback, 0, (size_t)nunits * sizeof(u64), ctx->stream));
    hipLaunchKernelGGL(k_compact_flags, dim3(std::min<u32>(nunits, (u32)ctx->cus * 8)), dim3(64), 0, ctx->stream, b->rflags, b->n, nunits,
                       ctx->d_ticket, ctx->d_lookback, b->subset);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    b->nsub = b->n_nonacgt;
    return BSK_OK;
}

template <class K>
static int blocks_per_cu(K kernel) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, 64, 0) != hipSuccess || nb < 1) {
        (void)hipGetLastError();
        nb = 1;
    }
    return nb;
}

// Which kernel runs a (batch, params) pair, on how many workgroups, and how the tuple arrays are organised.
enum Which { K_MIN_GEN_P, K_MIN_GEN_A, K_NT_P, K_NT_A, K_MIN_FAST, K_NT_FAST, K_SYN_P, K_SYN_A, K_KMER_P, K_KMER_A, K_SIM_P, K_SIM_A,
             K_PROT_HASH, K_PROT_MIN, K_SYN_FAST, K_PROT_MIN_FAST, K_PROT_HASH_FAST, K_SIM_FAST, K_MIN_DENSE, K_MIN_SEG, K_MIN_WPR, K_MIN_PK, K_SYN_PK, K_MIN_RING, K_SYN_SEL, K_MIN_PKD, K_MIN_DENSE_A, K_SYN_FAST_A };
struct Plan {
    Which which = K_MIN_GEN_P;
    int grid = 1;
    int fast_w = 0;
    bool syn_fused = false; // K_SYN_PK: k_syncmer_pf (the emit fused into every unit: no staging columns, kernels_syncmer_pf.hpp)
    bool syn_long = false;  // K_SYN_PK: k_syncmer_pkl (longer columns, more words in registers, two waves per SIMD)
    bool slab = false;     // true: unit u owns tuples [u*slab_unit, (u+1)*slab_unit) (+ overflow region); no look-back
    u64 slab_unit = 0;     // tuples per unit slab
    u64 slab_total = 0;    // nunits * slab_unit
    u32 nunits = 0;
    u32 ring_w = 0;
    size_t ring_entries = 0;
    u64 slab_read = 0;     // per-sequence slabs (protein fast path)
    int fast_k = 0;
    bool compact = false;  // stream kernels, fixed-length batch: runs without padding (k_nthash_fast<MODE, true>)
    u32 bin_gran = 0;      // != 0: the kernel runs over the batch's length-binned descriptors (ensure_binned), classes of this many bases
    bool fused_dna = false;  // protein minimizer of a 2-bit DNA batch: the kernel translates where it fetches its residues
    // mixed batch: the fast 2-bit kernel over all reads + the general ASCII kernel over the reads with a non-ACGT letter
    bool mixed = false;
    Which side_which = K_MIN_GEN_A;
    u32 side_nunits = 0, side_ring_w = 0;
    u64 side_slab = 0;     // K_MIN_DENSE_A: tuples of a read's slab in the side launch's region
    int side_grid = 1;
};

static_assert(sizeof(Plan) <= sizeof(((bsk_result *)nullptr)->plan_blob) && std::is_trivially_copyable<Plan>::value, "bsk_result::plan_blob holds a Plan");
static void plan_record(bsk_result *res, const bsk_batch *b, const bsk_params *p, int circ_ext, const Plan &pl) {
    memcpy(res->plan_blob, &pl, sizeof pl);
    res->plan_params = *p;
    res->plan_n = b->n;
    res->plan_bases = b->n_bases;
    res->plan_maxlen = b->maxlen;
    res->plan_circ = circ_ext;
    res->plan_valid = true;
}
static bool plan_recall(const bsk_result *res, const bsk_batch *b, const bsk_params *p, int circ_ext, Plan &pl) {
    if (!res->plan_valid || res->plan_n != b->n || res->plan_bases != b->n_bases || res->plan_maxlen != b->maxlen || res->plan_circ != circ_ext ||
        memcmp(&res->plan_params, p, sizeof *p) != 0)
        return false;
    memcpy(&pl, res->plan_blob, sizeof pl);
    return true;
}

// per-read slabs are sized by the LONGEST read: acceptable only while that does not blow the result arrays up (a batch of
// short reads with one long outlier would otherwise reserve the outlier's slab for every read)
static bool slab_budget_ok(const bsk_batch *b, u64 slab_read) {
    const double mean = b->n ? (double)b->n_bases / (double)b->n : 0.0;
    return (double)b->maxlen <= 4.0 * mean + 64.0 || (double)b->n * (double)slab_read * 12.0 < 256.0 * 1024 * 1024;
}

// Length binning pays when the reads of a unit end more than half a block of `step` k-mers apart (KArgs::binned, k_bin_desc): ragged
// batches of short reads on the lock-step kernels.  Returns the bases per length class (a multiple of step, at most 63 classes), 0: no.
static u32 bin_gran_for(const bsk_ctx *ctx, const bsk_batch *b, int step) {
    if (ctx->opt.no_bin || b->uniform_len || !b->desc || b->alias || b->maxlen >= 4096u || b->n < (u64)ctx->opt.bin_min || step < 1) return 0;
    const double mean = (double)b->n_bases / (double)b->n;
    if (((double)b->maxlen - mean) * 2.0 < (double)step) return 0;
    u32 g = (u32)step;
    while (b->maxlen / g > 61u) g += (u32)step;
    return g;
}
// the batch's length-binned descriptors for classes of `gran` bases above `lo`, built on the context's stream on first use and kept
// with the batch
static int ensure_binned(bsk_ctx *ctx, const bsk_batch *b, u32 lo, u32 gran, u32 mlo, u32 mhi, u32 mpretend, bool fine) {
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
static u64 ring_rows(double nwin, int w) {
    const double nw = std::max(nwin, 1.0);
    return ((u64)std::min(nw, std::ceil(nw * PlannerTable::slab_sel_num / (w + 1.0)) + 6.0) + 3) & ~(u64)3;
}

#define BSK_RESIZE (-1001)         // internal: a timed re-run outgrew the regions the result was sized with (run_planned_resizing sizes again, once)
#define BSK_REPLAN_CLASS (-1002)   // internal: a PART of a class plan overflowed on the launch the caller sees (run_classed sizes the parts again)
#define BSK_REPLAN_UNFUSED (-1000)  // internal: the fused DNA -> protein plan gave up, run the two-step path

static bool which_is_fast(Which w) {
    return w == K_MIN_FAST || w == K_NT_FAST || w == K_SYN_FAST || w == K_SIM_FAST || w == K_MIN_DENSE || w == K_MIN_SEG || w == K_MIN_WPR || w == K_MIN_PK || w == K_SYN_PK || w == K_MIN_RING || w == K_SYN_SEL || w == K_MIN_PKD;
}

static int make_plan_enc(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p, Plan &pl, bool use_ascii);

static int make_plan(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p, Plan &pl) {
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

static int make_plan_enc(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p, Plan &pl, bool use_ascii) {
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
            !ctx->no_syn_pk && slab_budget_ok(b, ring_rows(nwin, p->w))) {
            pl.which = K_MIN_RING;  // w <= 13: packed window machine, unit rows through a ring of staged rows (kernels_ring.hpp)
            pl.fast_w = p->w;
            pl.fast_k = b->maxlen > ring_minimizer_short_bases() ? 1 : 0;  // (the kernel's LONG argument, for plan_name)
            pl.slab = true;
            pl.slab_unit = (u64)64 * ring_rows(nwin, p->w);
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
            per_cu = use_ascii ? blocks_per_cu(k_minimizer_generic<1>) : blocks_per_cu(k_minimizer_generic<0>);
            pl.ring_w = (u32)p->w;
        }
    } else if (p->kind == BSK_NTHASH) {
        if (!use_ascii && b->maxlen + (u32)(p->circular ? p->k : 0) <= 16u * (BSK_NT_FAST_WORDS - 2) && !ctx->opt.force_generic) {
            pl.which = K_NT_FAST;
            // write-bound kernel: measured fastest at 4 waves/CU (more concurrent 128-byte write streams per XCD
            // thrash the L2 write-combining: 3.65 ms vs 5.27 ms at 15 waves/CU for 10M reads)
            pl.compact = stream_compact_ok(ctx, b, p);
            per_cu = p->canonical ? blocks_per_cu(k_nthash_fast<1>) : blocks_per_cu(k_nthash_fast<0>);
#ifdef BSK_EXPERIMENTS
            if (pl.compact) per_cu = p->canonical ? blocks_per_cu(k_nthash_fast<1, true>) : blocks_per_cu(k_nthash_fast<0, true>);
#endif
        } else {
            pl.which = use_ascii ? K_NT_A : K_NT_P;
            per_cu = use_ascii ? blocks_per_cu(k_nthash_stream<1>) : blocks_per_cu(k_nthash_stream<0>);
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
            per_cu = use_ascii ? blocks_per_cu(k_syncmer<1>) : blocks_per_cu(k_syncmer<0>);
            pl.ring_w = (u32)std::max(1, 2 * (p->k - p->s));
        }
    } else if (p->kind == BSK_KMER) {
        if (!use_ascii && p->canonical > 0 && b->maxlen + (u32)(p->circular ? p->k : 0) <= 16u * (BSK_NT_FAST_WORDS - 2) &&
            !ctx->opt.force_generic) {
            pl.which = K_NT_FAST;  // same streaming kernel, MODE 2
            pl.compact = stream_compact_ok(ctx, b, p);
            per_cu = blocks_per_cu(k_nthash_fast<2>);
#ifdef BSK_EXPERIMENTS
            if (pl.compact) per_cu = blocks_per_cu(k_nthash_fast<2, true>);
#endif
        } else {
            pl.which = use_ascii ? K_KMER_A : K_KMER_P;
            per_cu = use_ascii ? blocks_per_cu(k_kmer<1>) : blocks_per_cu(k_kmer<0>);
        }
    } else if (p->kind == BSK_SIMHASH) {
        const int nh = p->k - p->m + 1;
        if (!use_ascii && nh <= 63 && b->maxlen + (u32)(p->circular ? p->k : 0) <= 16u * (BSK_NT_FAST_WORDS - 2) &&
            !ctx->opt.force_generic) {
            pl.which = K_SIM_FAST;  // bit-sliced counters: 5 planes count to 31, 6 to 63
            pl.fast_w = nh <= 31 ? 5 : 6;
            const u32 ext_len = b->maxlen + (u32)(p->circular ? p->k : 0);
            pl.fast_k = ext_len <= 16u * (BSK_SIM_SHORT_WORDS - 2) ? 1 : ext_len <= 16u * (BSK_SIM_MID_WORDS - 2) ? 2 : 0;  // shorter reads: less LDS, more waves
            if (pl.fast_k == 1) per_cu = nh <= 31 ? blocks_per_cu(k_simhash_fast<5, BSK_SIM_SHORT_WORDS>) : blocks_per_cu(k_simhash_fast<6, BSK_SIM_SHORT_WORDS>);
            else if (pl.fast_k == 2) per_cu = nh <= 31 ? blocks_per_cu(k_simhash_fast<5, BSK_SIM_MID_WORDS>) : blocks_per_cu(k_simhash_fast<6, BSK_SIM_MID_WORDS>);
            else per_cu = nh <= 31 ? blocks_per_cu(k_simhash_fast<5>) : blocks_per_cu(k_simhash_fast<6>);
        } else {
            pl.which = use_ascii ? K_SIM_A : K_SIM_P;
            per_cu = use_ascii ? blocks_per_cu(k_simhash<1>) : blocks_per_cu(k_simhash<0>);
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
            per_cu = blocks_per_cu(k_prot_hash);
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
            per_cu = blocks_per_cu(k_prot_minimizer);
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
        int rc = ensure_scratch(ctx, nunits, 0);
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
            (e = hipMemsetAsync(ctx->d_lookback, 0, (size_t)nunits * sizeof(u64), ctx->stream)) == hipSuccess) {
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

// name of the planned kernel, as rocprofv3 shows it (bsk_result_plan; bench.py's roofline.kernel)
static void plan_name(const Plan &pl, const bsk_params *p, bool tiled, int cus, bsk_result *res) {
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
        case K_NT_FAST: snprintf(b, sizeof b, pl.compact ? "k_nthash_fast<%d,true>" : "k_nthash_fast<%d>", p->kind == BSK_KMER ? 2 : p->canonical ? 1 : 0); break;
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
static u64 syn_pk_fixcap(u64 n, int grid) { return std::max<u64>((u64)grid * 1024, (n / 4 + (u64)grid) / (u64)grid * (u64)grid); }

// ------------------------------------------------------------------------------------
// class plans: one plan per LENGTH CLASS of a batch instead of one plan per batch
// ------------------------------------------------------------------------------------
// The reference sketches one sequence at a time: a 5-kb contig costs 5 kb, whatever else is in the file (sketch.go:46, :85-94).  A batch
// plan keyed on the longest read does not: one 400-base read moved 10^8 x 150 bases from k_minimizer_pk to k_minimizer_dense, one 5-kb
// read moved them onto tiles.  A class plan cuts the batch by length at the points where the planner's choice changes (LenHist: known on
// the host since the batch was created), runs the BULK class -- the one with most bases -- over a view of the batch in which every other
// read has length 0, and every other class as a batch of its own (its descriptors gathered, the words shared) whose slabs live in the
// TAIL of the parent's arrays; k_adopt_refs then points those reads' reference words there.  Callers see one result.
struct ClassPart {
    bsk_batch *sub = nullptr;   // borrowed view: desc = the class's descriptors (context pool), words = the parent's
    bsk_result *res = nullptr;  // refs / status of its own; hash / pos = the parent's tail once the parent exists
    u32 *list = nullptr;        // the class's reads (batch positions), ascending (context pool)
    u64 n = 0, bases = 0, off = 0, extent = 0;
    u32 lo = 0, hi = 0;
    bool tiled = false;  // longer than the kind's tile threshold: the part runs over tiles (sketch_tiled), its result is wide and copied into the tail
    bool fresh = false;  // ... and was just run by the sizing call (the parent's first launch does not run it again)
    bool async = false;  // ... without a synchronisation of its own (sketch_tiled, tile_async): its overflow flags wait in the side context's d_ticket[24]
};
struct ClassSet {
    std::vector<ClassPart> parts;
    bsk_batch *view = nullptr;  // the bulk class's view of the batch
    u64 tail = 0;
    u64 n = 0, n_bases = 0;     // what it was cut from
    u32 maxlen = 0, blo = 0, bhi = 0;
    const u64 *desc = nullptr;
    const u32 *words = nullptr;
    float build_ms = 0.0f;
    bool masked = false;   // no view array: the bulk's kernel masks by length itself (KArgs::cls_lo / cls_hi / cls_pretend)
    u32 pretend = 0;
};
static void class_set_free(ClassSet *cs) {
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

static int launch(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p, bsk_result *res, int circ_ext, const Plan &pl, hipEvent_t ev0, hipEvent_t ev1);
// the parts of a class plan into the tail of `res` (called from the parent's launch, before its own kernel)
static int sketch_tiled(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p_in, int circ_ext, bsk_result **result, int warmup, int iters, float *kernel_ms);
// The parts run on the side context's stream (plain launches over batches of their own; their slabs are slices of the parent's tail):
// tiled = false: the parts that are one launch each, queued BEFORE the bulk's kernel so that they take their few CU slots first and
// the bulk's persistent waves fill the rest (the parts' latency-bound launches then overlap with the bulk); tiled = true: the parts that
// run over tiles (sketch_tiled: several kernels and host round trips) -- first of all.
static int launch_parts(bsk_ctx *ctx, ClassSet *cs, const bsk_params *p, bsk_result *res, bool tiled) {
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
static int adopt_parts(bsk_ctx *ctx, ClassSet *cs, bsk_result *res) {
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

// One launch of the planned kernel into res.  ev0/ev1 (optional) bracket the kernel itself (with the parts of a class plan).
static int launch(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p, bsk_result *res, int circ_ext, const Plan &pl,
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
            if (a.kind == BSK_KMER) hipLaunchKernelGGL(k_nthash_fast<2>, dim3(pl.grid), dim3(64), 0, ctx->stream, a);
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
static u64 estimate_cap_n(const bsk_params *p, u64 bases, u64 nreads);
static u64 estimate_cap(const bsk_batch *b, const bsk_params *p, int circ_ext) {
    if (p->kind == BSK_PROT_HASH && b->alphabet == BSK_ALPHA_DNA) return estimate_cap_n(p, b->n_bases / 3 + b->n, b->n);  // fused: residues
    return estimate_cap_n(p, b->n_bases + b->n * (u64)circ_ext, b->n);
}
static u64 estimate_cap_n(const bsk_params *p, u64 bases, u64 nreads) {
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

static int make_circular(bsk_ctx *ctx, const bsk_batch *b, int k, bsk_batch **out);

// Plan, size, launch (and optionally time) the kernel of p->kind over a prepared batch.
static int run_planned(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p, int circ_ext, bsk_result **result, int warmup, int iters,
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
        ovf_cap += (b->n / 64 + 64) * ((nwin_max + 15) & ~(u64)15);
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
static int run_planned_resizing(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p, int circ_ext, bsk_result **result, int warmup, int iters,
                                float *kernel_ms) {
    int rc = run_planned(ctx, b, p, circ_ext, result, warmup, iters, kernel_ms);
    if (rc != BSK_RESIZE) return rc;
    ctx->in_resize = true;
    rc = run_planned(ctx, b, p, circ_ext, result, 0, 0, nullptr);
    if (rc == BSK_OK) rc = run_planned(ctx, b, p, circ_ext, result, warmup, iters, kernel_ms);
    ctx->in_resize = false;
    return rc;
}

// ------------------------------------------------------------------------------------
// long sequences: tile, run the ordinary kernels over the tiles, stitch (kernels_tile.hpp)
// ------------------------------------------------------------------------------------
static bool kind_tiles(const bsk_params *p) {
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

static int sketch_tiled(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p_in, int circ_ext, bsk_result **result, int warmup, int iters,
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
            TCHK(hipMalloc(&fin->hash, need * 8));
            TCHK(hipMalloc(&fin->pos, need * 4));
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
        TCHK(hipMalloc(&fin->hash, fin->cap * 8));
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
            TCHK(hipMalloc(&fin->hash, cap * 8));
            TCHK(hipMalloc(&fin->pos, cap * 4));
            fin->cap = fin->alloc_cap = cap;
        }
        lap("result arrays");
        TCHK(pool(9, (nt + 1) * 8, (void **)&oexcl));
        TCHK(hipMemsetAsync(oexcl, 0, (nt + 1) * 8, ctx->stream));
        if (nt) {
            const u32 tunits = (u32)((nt + 63) / 64);
            rc = ensure_scratch(ctx, stitch_scratch_words(tunits), 0);
            if (rc != BSK_OK) return done(rc);
            TCHK(hipMemsetAsync(ctx->d_ticket, 0, 2 * sizeof(u32), ctx->stream));
            TCHK(hipMemsetAsync(ctx->d_lookback, 0, stitch_scratch_words(tunits) * 8, ctx->stream));
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
static u32 tile_min_for(const bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p) {
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

// ---- class plans: the decision (host, from the batch's length histogram) ------------------------------------------------------------
struct ClassSig {
    int which = -1, octave = 0;
    bool syn_long = false, syn_fused = false;
    bool operator==(const ClassSig &o) const { return which == o.which && octave == o.octave && syn_long == o.syn_long && syn_fused == o.syn_fused; }
};
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
struct ClassCut {
    u32 lo, hi;   // the class takes the lengths [lo, hi]
    u32 shortest; // the shortest read it holds
    u64 n, bases;
    ClassSig sig;
};
// -> the classes (ascending) and the index of the bulk; false: keep one plan
static bool class_decide(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p, int circ_ext, std::vector<ClassCut> &cuts, int &bulk) {
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

static int run_planned(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p, int circ_ext, bsk_result **result, int warmup, int iters, float *kernel_ms);
static int run_planned_resizing(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p, int circ_ext, bsk_result **result, int warmup, int iters, float *kernel_ms);

// lists, views and sub-batches of the classes (device passes on the context's stream; the arrays live in the context's pool)
static int class_build(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p, const std::vector<ClassCut> &cuts, int bulk, ClassSet *cs) {
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
static int run_classed(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p, int circ_ext, bsk_result **result, int warmup, int iters, float *kernel_ms, bool *applied) {
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
static int make_circular(bsk_ctx *ctx, const bsk_batch *b, int k, bsk_batch **out) {
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
