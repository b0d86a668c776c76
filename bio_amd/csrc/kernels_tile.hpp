// kernels_tile.hpp -- long sequences (contigs, chromosomes) as overlapping tiles.
//
// The sketch kernels give one sequence to one lane; that is the right shape for reads, not for a 100-Mbp chromosome.
// A long sequence is therefore cut into tiles that the same kernels process as ordinary short "reads" of one batch
// (zero copy: a tile is a descriptor into the sequence's 2-bit words / ASCII bytes), and a stitch pass turns the tile
// results back into one result per sequence.  Exactness rests on the closed forms (note V):
//   * stream kinds (ntHash, k-mer code, SimHash): value i depends on bases [i, i+k-1] only.  Tile j owns values
//     [j*TP, (j+1)*TP) and reads TP+k-1 bases; TP is a multiple of 16, so tile starts are word aligned and the
//     line-padded runs of consecutive tiles are adjacent in the output: no copy at all.
//   * minimizer: the emitted tuples are the distinct values of p_j = leftmost argmin of window j, and p_j is in
//     [j, j+w-1].  Tile j owns POSITIONS [P0,P1): it evaluates windows max(0,P0-w+1) .. min(P1-1, nwin-1), which is every
//     window that can select a position of its range, and the stitch keeps the tuples whose position is in [P0,P1).
//   * syncmer (s < k): the emitted positions are the distinct b(idx) <= end, with b(idx) in [idx, idx+w-1], w = k-s.
//     Tile j evaluates idx = max(0,P0-w+1) .. P1-1; its own "end" rule drops b > P1-1, which belong to the next tile,
//     and in the last tile P1-1 = end is the reference's bound (sketch.go:170,314).
// A tile starts at a multiple of 16 bases (a0); positions are shifted back by a0 when stitching.
#pragma once
#include "biosketch.h"
#include "device_common.hpp"

namespace bsk {

struct TileGeo {
    int kind, k, w, s;  // w: minimizer window, or k - s for the syncmer
    u32 tp;             // positions owned by one tile (multiple of 16)
    int circ_ext;       // bases the circular copy appended (length checks look at L - circ_ext)
    int syn_all;        // the w = 1 minimizer stands in for a syncmer sketch with s == k: that constructor's length rule applies
};

struct SeqTab {  // where the sequences of a batch are: DNA desc (reads shorter than 2^24) or fw + llen; protein: aoff only
    const u64 *desc, *fw, *llen, *aoff;
    u64 n;
    const u8 *short_flag;  // protein batch that came out of the translate kernel: 1 = too few nucleotides (ErrShortSeq)
};
__device__ __forceinline__ u64 seq_first_word(const SeqTab &t, u64 r) { return t.desc ? t.desc[r] >> 24 : (t.fw ? t.fw[r] : 0); }
__device__ __forceinline__ u64 seq_len(const SeqTab &t, u64 r) {
    return t.desc ? t.desc[r] & 0xffffffULL : (t.llen ? t.llen[r] : t.aoff[r + 1] - t.aoff[r]);
}

// positions (values, or selectable k-mer starts) of a sequence of L bases; 0: the constructor returns ErrShortSeq
__host__ __device__ __forceinline__ u64 tile_npos(const TileGeo &g, u64 L, int short_flag = -1) {
    if (g.kind == BSK_PROT_HASH || g.kind == BSK_PROT_MINIMIZER) {
        // iterator-protein.go:50, sketch-protein.go:66,73: the constructors look at the INPUT length (3k, 3k+w-1); the
        // residues may then be too few for a k-mer / a window, which yields nothing
        const u64 wm1 = g.kind == BSK_PROT_MINIMIZER ? (u64)g.w - 1 : 0;
        const bool shrt = short_flag >= 0 ? short_flag != 0 : L < 3ULL * g.k + wm1;
        return (!shrt && L >= (u64)g.k + wm1) ? L - (u64)g.k + 1 : 0;
    }
    if (L < (u64)g.circ_ext) return 0;
    const u64 L0 = L - (u64)g.circ_ext;
    switch (g.kind) {
        case BSK_MINIMIZER:
            // (s == k syncmer: len(S.Seq) < 2k-s-1 = k-1 refuses, sketch.go:149, and the hasher needs k bases of the extended copy, :179)
            if (g.syn_all) return (L0 + 1 >= (u64)g.k && L >= (u64)g.k) ? L - (u64)g.k + 1 : 0;
            return (L0 + 1 >= (u64)g.k + (u64)g.w) ? L - (u64)g.k + 1 : 0;  // sketch.go:92
        case BSK_SYNCMER: return (L0 + 1 + (u64)g.s >= 2ULL * g.k && L >= (u64)g.k) ? L + (u64)g.s + 2 - 2ULL * g.k : 0;  // sketch.go:149,170
        default: return L0 >= (u64)g.k ? L - (u64)g.k + 1 : 0;  // iterator.go:619,672,128
    }
}
// does the reference constructor refuse the sequence (ErrShortSeq)?  Not the same as "no position": a translation can pass
// the nucleotide-length rule and still be too short for a k-mer or a window -- that yields nothing and is no error
__host__ __device__ __forceinline__ bool tile_short(const TileGeo &g, u64 L, int short_flag = -1) {
    if (g.kind == BSK_PROT_HASH || g.kind == BSK_PROT_MINIMIZER) {
        const u64 wm1 = g.kind == BSK_PROT_MINIMIZER ? (u64)g.w - 1 : 0;
        return short_flag >= 0 ? short_flag != 0 : L < 3ULL * g.k + wm1;
    }
    return tile_npos(g, L) == 0;
}
__host__ __device__ __forceinline__ u64 tile_count(const TileGeo &g, u64 L, int short_flag = -1) {
    return (tile_npos(g, L, short_flag) + g.tp - 1) / g.tp;
}
__device__ __forceinline__ int seq_short(const SeqTab &t, u64 r) { return t.short_flag ? (int)t.short_flag[r] : -1; }

struct TileArgs {
    SeqTab seq;
    TileGeo geo;
    u32 nunits;       // ceil(n / 64)
    u64 *tstart;      // [n+1] first tile of every sequence
    u32 *ticket;
    u64 *lookback;
};

// first tile of every sequence: exclusive prefix of the tile counts (one wavefront per 64 sequences + look-back)
__global__ __launch_bounds__(64) void k_tile_count(TileArgs a) {
    const int lane = lane_id();
    for (;;) {
        const u32 unit = next_ticket(a.ticket, lane);
        if (unit >= a.nunits) break;
        const u64 r = (u64)unit * 64 + lane;
        const u64 c = r < a.seq.n ? tile_count(a.geo, seq_len(a.seq, r), seq_short(a.seq, r)) : 0;
        const u64 incl = wave_incl_scan_u64(c, lane);
        const u64 base = lookback_exclusive(a.lookback, unit, wave_bcast_u64(incl, 63), lane);
        if (r < a.seq.n) a.tstart[r] = base + incl - c;
        if (unit == a.nunits - 1 && lane == 63) a.tstart[a.seq.n] = base + incl;
    }
}

struct TileTab {
    u64 *desc;    // [nt] (first_word << 24) | n_bases         (2-bit kernels)
    u64 *adesc;   // [nt] (first_byte << 24) | n_bases or NULL (ASCII kernels)
    u32 *seq;     // [nt] owning sequence
    u64 *shift;   // [nt] a0: tile-local position + a0 = position in the sequence
    u64 *keep;    // [nt] lo | hi << 32: tile-local positions [lo, hi) belong to this tile
};

// nt may be an UPPER BOUND of the number of tiles (the path without host round trips sizes everything from n_bases / tp + n): the entries
// from the true count tstart[n] on are empty tiles -- no bases, no owned positions -- that every later pass takes as such
__global__ void k_tile_build(SeqTab seq, TileGeo g, const u64 *tstart, u64 nt, TileTab t, const u32 *wbits, u8 *tflags) {
    const u64 nt_true = tstart[seq.n];
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < nt; i += (u64)gridDim.x * blockDim.x) {
        if (i >= nt_true) {
            t.desc[i] = 0;
            if (t.adesc) t.adesc[i] = 0;
            t.seq[i] = 0;
            t.shift[i] = 0;
            t.keep[i] = 0;
            if (tflags) tflags[i] = 0;
            continue;
        }
        u64 lo = 0, hi = seq.n - 1;  // largest r with tstart[r] <= i (sequences without tiles share their successor's start)
        while (lo < hi) {
            const u64 mid = (lo + hi + 1) >> 1;
            if (tstart[mid] <= i) lo = mid;
            else hi = mid - 1;
        }
        const u64 r = lo, L = seq_len(seq, r), np = tile_npos(g, L, seq_short(seq, r));
        const bool prot = g.kind == BSK_PROT_HASH || g.kind == BSK_PROT_MINIMIZER;
        const u64 P0 = (i - tstart[r]) * g.tp, P1 = (P0 + g.tp < np) ? P0 + g.tp : np;
        u64 a0, Lt;
        if (g.kind == BSK_MINIMIZER || g.kind == BSK_PROT_MINIMIZER) {
            const u64 nwin = np - (u64)g.w + 1;
            const u64 jlo = P0 + 1 > (u64)g.w ? P0 + 1 - (u64)g.w : 0;
            const u64 jhi = P1 - 1 < nwin - 1 ? P1 - 1 : nwin - 1;
            a0 = prot ? jlo : (jlo & ~15ULL);  // residues are bytes: no word alignment
            Lt = jhi + (u64)g.w + (u64)g.k - 1 - a0;
        } else if (g.kind == BSK_SYNCMER) {
            const u64 ilo = P0 + 1 > (u64)g.w ? P0 + 1 - (u64)g.w : 0;
            a0 = ilo & ~15ULL;
            Lt = (P1 - 1) + 2ULL * g.k - (u64)g.s - 1 - a0;
        } else {
            a0 = P0;
            Lt = (P1 - P0) + (u64)g.k - 1;
        }
        t.desc[i] = prot ? 0 : ((seq_first_word(seq, r) + (a0 >> 4)) << 24) | Lt;
        if (t.adesc) t.adesc[i] = ((seq.aoff[r] + a0) << 24) | Lt;
        t.seq[i] = (u32)r;
        t.shift[i] = a0;
        t.keep[i] = (P0 - a0) | ((P1 - a0) << 32);
        if (tflags) {  // any packed word of the tile marked by k_pack?
            const u64 w0 = seq_first_word(seq, r) + (a0 >> 4), w1 = w0 + ((Lt + 15) >> 4);
            u32 any = 0;
            for (u64 w = w0; w < w1; ++w) any |= (wbits[w >> 5] >> (w & 31)) & 1u;
            tflags[i] = any ? (u8)BSK_ST_HAS_NON_ACGT : (u8)0;
        }
    }
}

// per-tile flags -> per-sequence accumulators: first-window tie (first tile only), first tile that met an illegal base
__global__ void k_tile_flags(const u8 *tstatus, const u32 *tseq, const u64 *tstart, u64 nt, u64 nseq, u32 *sflags, u64 *sbad) {
    const u64 nt_true = tstart[nseq];  // (nt may be an upper bound: k_tile_build)
    if (nt > nt_true) nt = nt_true;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < nt; i += (u64)gridDim.x * blockDim.x) {
        const u8 st = tstatus[i];
        const u32 r = tseq[i];
        if ((st & BSK_ST_FIRST_WINDOW_TIE) && tstart[r] == i) atomicOr(&sflags[r], (u32)BSK_ST_FIRST_WINDOW_TIE);
        if ((st & BSK_ST_CODE_MASK) == BSK_ST_ILLEGAL) atomicMin((unsigned long long *)&sbad[r], (unsigned long long)i);
    }
}

struct StitchArgs {
    u64 nt;
    u32 nunits;          // ceil(nt / 64)
    const u64 *trefs;    // tile results
    u64 tcap;            // tuples the tile result's arrays hold: a reference word that names more (a launch that overflowed and was not sized again) counts as empty and raises the flag
    const u64 *thash;
    const u32 *tpos;
    const u64 *shift, *keep;
    u64 *oexcl;          // [nt+1] first output tuple of every tile
    u64 *ohash;          // stitched tuples
    u32 *opos;
    u64 cap;             // capacity of ohash / opos
    u32 *ticket;         // [1] = overflow flag
    u64 *lookback;
};

// minimizer / syncmer: keep the tuples a tile owns, shift their positions, pack them in tile order.
// A unit is 64 tiles.  Positions ascend inside a tile, so the tuples a tile OWNS (positions in [lo, hi): TileTab::keep) are ONE RUN of its
// tuples: every lane finds its tile's run by two binary searches over the tile's positions (both at once: ~log2(count) dependent loads, 5
// for a pk-sized tile, 8 for a pkd-sized one) -- that is the whole count pass; the unit's total goes through the look-back; then the runs
// are copied behind one another, 64 consecutive output tuples per step and four steps' loads in flight, no test per tuple: entry j of the
// unit's kept tuples belongs to the first tile whose inclusive kept-count prefix exceeds j (a table written by the tiles when the unit keeps
// at most OWN_CAP tuples; larger units -- pkd-sized tiles -- go tile by tile, a tile's run being several steps long there).
// (Round 4 gave every LANE a tile and scanned its ~21 positions one dependent load after the other: 6.1 ms for 4 10^8 tile tuples.  Round 5
// streamed all entries twice -- count pass, write pass, a test per entry -- and searched the prefix per entry in units above OWN_CAP: six
// dependent LDS round trips per entry, 2.7 ms for the pkd-sized tiles of 2 10^9 bases, 4.2 TB/s of traffic at best.)
// (scratch of a stitch over nunits units: lb_words_with_heads -- the look-back granules, then the ticket heads)
__global__ __launch_bounds__(64) void k_tile_stitch(StitchArgs a) {
    constexpr u32 OWN_CAP = 4096;  // kept tuples whose tile is looked up in a table (a unit of pk tiles keeps ~1 500)
    constexpr int SU = 4;
    __shared__ u64 s_src[64], s_shift[64];  // first kept tuple of the tile, its offset in the sequence
    __shared__ u32 s_pre[64], s_stride[64];
    __shared__ u8 s_own[OWN_CAP];
    const int lane = lane_id();
    HeadTickets tickets(reinterpret_cast<u32 *>(a.lookback + lb_heads_at(a.nunits)));  // (one unit per ticket: device_common.hpp, "tickets from eight heads")
    for (;;) {
        const u32 unit = tickets.next(a.nunits, lane);
        if (unit == ~0u) break;
        const u64 t = (u64)unit * 64 + lane;
        u32 kept = 0;
        {
            u64 first = 0, sh = 0;
            u32 cnt = 0, lo = 0, hi = 0, stride = 1;
            if (t < a.nt) {
                const u64 ref = a.trefs[t], kp = a.keep[t];
                first = BSK_REF_FIRST(ref);
                cnt = (u32)(ref & 0xffffffULL);
                stride = (u32)BSK_REF_STRIDE(ref);
                if (cnt && first + (u64)(cnt - 1u) * stride >= a.tcap) {
                    cnt = 0;
                    atomicOr(&a.ticket[1], 1u);
                }
                lo = (u32)kp;
                hi = (u32)(kp >> 32);
                sh = a.shift[t];
            }
            // first tuple at or beyond lo, first tuple at or beyond hi
            u32 a0 = 0, b0 = cnt, a1 = 0, b1 = cnt;
            while (__builtin_amdgcn_ballot_w64(a0 < b0 || a1 < b1)) {
                const u32 m0 = (a0 + b0) >> 1, m1 = (a1 + b1) >> 1;
                const bool g0 = a0 < b0, g1 = a1 < b1;
                const u32 p0 = g0 ? a.tpos[first + (u64)m0 * stride] & BSK_POS_MASK : 0u;
                const u32 p1 = g1 ? a.tpos[first + (u64)m1 * stride] & BSK_POS_MASK : 0u;
                if (g0) {
                    if (p0 < lo) a0 = m0 + 1u;
                    else b0 = m0;
                }
                if (g1) {
                    if (p1 < hi) a1 = m1 + 1u;
                    else b1 = m1;
                }
            }
            kept = a1 > a0 ? a1 - a0 : 0u;
            s_src[lane] = first + (u64)a0 * stride;
            s_shift[lane] = sh;
            s_stride[lane] = stride;
        }
        const u32 pre = wave_incl_scan_u32(kept, lane);
        s_pre[lane] = pre;
        const u32 K = wave_bcast_u32(pre, 63);
        const u32 start = pre - kept;  // this lane's TILE starts at kept tuple `start` of the unit
        const u64 base = lookback_exclusive(a.lookback, unit, (u64)K, lane);
        if (t < a.nt) a.oexcl[t] = base + start;
        if (unit == a.nunits - 1 && lane == 63) a.oexcl[a.nt] = base + K;
        if (base + K > a.cap) {
            if (lane == 0) atomicOr(&a.ticket[1], 1u);
            wave_sync_lds();
            continue;
        }
        if (K <= OWN_CAP) {
            // the table tuple -> tile: every tile writes its number over its own run
            for (u32 i = 0; i < kept; ++i) s_own[start + i] = (u8)lane;
            wave_sync_lds();
            for (u32 j0 = 0; j0 < K; j0 += 64 * SU) {
                u32 pv[SU], ov[SU];
                u64 hv[SU];
#pragma unroll
                for (int q = 0; q < SU; ++q) {
                    const u32 j = j0 + 64u * (u32)q + (u32)lane;
                    pv[q] = 0;
                    ov[q] = 0;
                    hv[q] = 0;
                    if (j < K) {
                        const u32 o = s_own[j];
                        const u64 src = s_src[o] + (u64)(j - (o ? s_pre[o - 1] : 0u)) * s_stride[o];  // (s_pre is inclusive)
                        ov[q] = o;
                        pv[q] = __builtin_nontemporal_load(&a.tpos[src]);
                        hv[q] = __builtin_nontemporal_load(&a.thash[src]);
                    }
                }
#pragma unroll
                for (int q = 0; q < SU; ++q) {
                    const u32 j = j0 + 64u * (u32)q + (u32)lane;
                    if (j < K) {
                        __builtin_nontemporal_store(hv[q], &a.ohash[base + j]);
                        __builtin_nontemporal_store(pv[q] + (u32)s_shift[ov[q]], &a.opos[base + j]);  // (the strand bit rides along: a sequence position stays below 2^31)
                    }
                }
            }
        } else {
            wave_sync_lds();
            for (u32 o = 0; o < 64u; ++o) {  // tile by tile (a pkd-sized tile keeps ~160 tuples: one step)
                const u32 before = o ? s_pre[o - 1] : 0u, n_o = s_pre[o] - before;
                const u64 src0 = s_src[o], dst0 = base + before;
                const u32 stride = s_stride[o], sh = (u32)s_shift[o];
                for (u32 i0 = 0; i0 < n_o; i0 += 64 * SU) {
                    u32 pv[SU];
                    u64 hv[SU];
#pragma unroll
                    for (int q = 0; q < SU; ++q) {
                        const u32 i = i0 + 64u * (u32)q + (u32)lane;
                        pv[q] = 0;
                        hv[q] = 0;
                        if (i < n_o) {
                            pv[q] = __builtin_nontemporal_load(&a.tpos[src0 + (u64)i * stride]);
                            hv[q] = __builtin_nontemporal_load(&a.thash[src0 + (u64)i * stride]);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < SU; ++q) {
                        const u32 i = i0 + 64u * (u32)q + (u32)lane;
                        if (i < n_o) {
                            __builtin_nontemporal_store(hv[q], &a.ohash[dst0 + i]);
                            __builtin_nontemporal_store(pv[q] + sh, &a.opos[dst0 + i]);
                        }
                    }
                }
            }
        }
        wave_sync_lds();
    }
}

// per sequence: first tuple, tuple count, status.  stream = 1: the tile runs already are the sequence's run.
// dense = 1 (k_minimizer_pft): the tile kernel packed its tiles' OWNED tuples back to back in tile order -- a sequence's tuples are the run from its
// first tile's first tuple to the end of its last tile's.
__global__ void k_tile_finish(SeqTab seq, TileGeo g, const u64 *tstart, const u64 *oexcl /*stitched kinds*/, const u64 *trefs /*stream kinds, dense tiles*/,
                              const u8 *rflags, const u32 *sflags, const u64 *sbad, u64 *wfirst, u64 *wcount, u8 *status, u64 *total, int dense = 0) {
    u64 sum = 0;
    for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < seq.n; r += (u64)gridDim.x * blockDim.x) {
        const u64 t0 = tstart[r], t1 = tstart[r + 1];
        u64 first = 0, cnt = 0;
        u8 st = tile_short(g, seq_len(seq, r), seq_short(seq, r)) ? BSK_ST_SHORT : BSK_ST_OK;
        if (t1 > t0) {
            if (dense) {
                const u64 last = trefs[t1 - 1];
                first = trefs[t0] >> 24;
                cnt = (last >> 24) + (last & 0xffffffULL) - first;
            } else if (oexcl) {
                first = oexcl[t0];
                cnt = oexcl[t1] - first;
            } else {
                first = trefs[t0] >> 24;
                cnt = tile_npos(g, seq_len(seq, r), seq_short(seq, r));
                const u64 bad = sbad[r];
                if (bad != ~0ULL) {  // NextKmer stopped at the first k-mer with an illegal base (iterator.go:746)
                    cnt = (bad - t0) * g.tp + (trefs[bad] & 0xffffffULL);
                    st = BSK_ST_ILLEGAL;
                }
            }
            st |= (u8)sflags[r];
            if (rflags) st |= rflags[r];
        }
        wfirst[r] = first;
        wcount[r] = cnt;
        status[r] = st;
        sum += cnt;
    }
    sum = wave_sum_u64(sum);
    if ((threadIdx.x & 63) == 0 && sum) atomicAdd((unsigned long long *)total, (unsigned long long)sum);
}

// Two-strand k-mer codes over tiles (NextKmer with canonical = false, iterator.go:713-723): once the forward strand is exhausted
// the reference reverse-complements the sequence in place and walks it again, Index() restarting at 0 -- value j of the second
// strand is the reverse complement of forward k-mer n-1-j.  A sequence that hit an illegal base never gets there
// (iterator.go:746-748 returns the error first).  The forward codes of sequence r sit at src[first, first + cnt); its two strands
// go to dst[2 first, 2 first + 2 cnt).  One workgroup per tile.
// A sequence with a letter outside ACGTacgt takes its second strand from the bytes: RevComInplace pairs LETTERS (N stays N, R <-> Y;
// seq/alphabet.go:361-367) and base2bit then maps the paired letter, which is not the complement of the code.
__global__ __launch_bounds__(256) void k_two_strand(const u64 *trefs, const u32 *tseq, u64 nt, const u64 *src, u64 *dst, const u64 *wfirst,
                                                    const u64 *wcount, const u8 *status, int k, const u8 *ascii, const u64 *aoff, int pairs) {
    const unsigned map2 = pair_map2(pairs);
    const unsigned sh = 64u - 2u * (unsigned)k;
    for (u64 t = blockIdx.x; t < nt; t += gridDim.x) {
        const u64 ref = trefs[t];
        const u64 base = ref >> 24;
        const u32 cnt_t = (u32)(ref & 0xffffffULL);
        const u32 r = tseq[t];
        const u64 first = wfirst[r], cnt = wcount[r];
        const bool two = (status[r] & BSK_ST_CODE_MASK) != BSK_ST_ILLEGAL;
        const u8 *letters = (ascii && (status[r] & BSK_ST_HAS_NON_ACGT)) ? ascii + aoff[r] : nullptr;
        for (u32 i = threadIdx.x; i < cnt_t; i += blockDim.x) {
            const u64 j = base + i - first;
            if (j >= cnt) break;
            const u64 f = src[base + i];
            dst[2 * first + j] = f;
            if (two && letters) {
                u64 v = 0;  // first base of the second-strand k-mer = pair of the last letter of forward k-mer j
                for (int t = k - 1; t >= 0; --t) v = (v << 2) | (base2bit_dev(dna_pair_dev(letters[j + (u64)t], pairs)) & 3u);
                dst[2 * first + 2 * cnt - 1 - j] = v;
            } else if (two && map2 != 0x1Bu) {  // RNA alphabets keep a 'T', Unlimit complements nothing: base by base
                u64 v = 0;
                for (int t = 0; t < k; ++t) v = (v << 2) | ((map2 >> (2u * (unsigned)((f >> (2 * t)) & 3u))) & 3u);
                dst[2 * first + 2 * cnt - 1 - j] = v;
            } else if (two) {
                u64 v = ~f;  // complement, then reverse the order of the 2-bit groups
                v = ((v >> 2) & 0x3333333333333333ULL) | ((v & 0x3333333333333333ULL) << 2);
                v = ((v >> 4) & 0x0f0f0f0f0f0f0f0fULL) | ((v & 0x0f0f0f0f0f0f0f0fULL) << 4);
                v = __builtin_bswap64(v);
                dst[2 * first + 2 * cnt - 1 - j] = v >> sh;
            }
        }
    }
}

__global__ void k_two_strand_refs(u64 n, u64 *wfirst, u64 *wcount, const u8 *status, u64 *total) {
    u64 sum = 0;
    for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (u64)gridDim.x * blockDim.x) {
        const u64 cnt = wcount[r];
        wfirst[r] *= 2;
        if ((status[r] & BSK_ST_CODE_MASK) != BSK_ST_ILLEGAL) {
            wcount[r] = 2 * cnt;
            sum += cnt;
        }
    }
    sum = wave_sum_u64(sum);
    if ((threadIdx.x & 63) == 0 && sum) atomicAdd((unsigned long long *)total, (unsigned long long)sum);
}

}  // namespace bsk
