// kernels_fast.hpp -- minimizer kernel specialised on the window size W (compile time).
//
// Same mapping as kernels_generic.hpp (one read per lane, 64 reads per wavefront,
// wave-uniform loop counters) but the k-mer loop is unrolled by W so that
//   * the sliding-window state (suffix minima of the previous block of W k-mers,
//     running prefix minimum of the current block) lives entirely in VGPRs with
//     static indices -- no LDS/HBM traffic for the window at all;
//   * the 2-bit codes of the W incoming and W outgoing bases of a block are cut
//     out of the packed stream once per block (one v_alignbit each) and then
//     addressed with immediate shifts; the packed words of the NEXT block are
//     requested while the current block is being hashed;
//   * per k-mer the only memory read is ONE 16-byte LDS fetch of the
//     (outgoing, incoming) -> (forward, reverse) update table, and all W of them
//     are issued at the top of the block, ahead of the dependent hash chain;
//   * the block body is branch-free: a selected tuple is stored to the lane's next
//     LDS staging slot, a non-selected one to a per-lane dummy slot.
// HBM traffic is the packed bases in and the selected (hash, pos|strand) tuples
// out; tuples leave LDS as whole 512-/256-byte rows.
#pragma once
#include "kernels_generic.hpp"

namespace bsk {

// LDS staging for the fast kernels: slot(e, lane) = e*65 + lane (row stride 65: a lane's
// consecutive tuples rotate through the banks; lanes at equal e are consecutive).
// Row CAP is the dummy row (slot CAP*65 + lane) that absorbs non-selected stores.
template <int CAP>
struct FStage {
    u64 *sh;  // [(CAP+1)*65]
    u32 *sp;  // [(CAP+1)*65]
    u16 *smap;
    static constexpr int SLOTS = (CAP + 1) * 65;
};

struct H64 {
    u32 lo, hi;
};
__device__ __forceinline__ u64 to64(H64 v) { return ((u64)v.hi << 32) | v.lo; }
__device__ __forceinline__ H64 hrol1(H64 v) {
    H64 r;
    r.lo = __builtin_amdgcn_alignbit(v.lo, v.hi, 31);
    r.hi = __builtin_amdgcn_alignbit(v.hi, v.lo, 31);
    return r;
}
__device__ __forceinline__ H64 hror1(H64 v) {
    H64 r;
    r.lo = __builtin_amdgcn_alignbit(v.hi, v.lo, 1);
    r.hi = __builtin_amdgcn_alignbit(v.lo, v.hi, 1);
    return r;
}

template <int W, int CAP, bool DIRECT>
struct FastMin {
    const u32 *__restrict__ w;
    const uint4 *__restrict__ xt;
    int k, lane;
    u32 nk;
    FStage<CAP> st;
    u64 *__restrict__ ghash;
    u32 *__restrict__ gpos;
    u64 gbase;
    // rolling state
    H64 fh, rh;
    u64 bh[W];
    u32 bp[W];
    u64 Ph;
    u32 Pp, prev, cnt, tie;
    // packed words of the current block: (in_lo,in_hi) from base t0, (out_lo,out_hi) from base i0-1
    u32 in_lo, in_hi, out_lo, out_hi;

    __device__ __forceinline__ void load_block_words(u32 i0) {
        const u32 t0 = i0 + (u32)k - 1;
        in_lo = w[t0 >> 4];
        in_hi = w[(t0 >> 4) + 1];
        const u32 p0 = i0 ? i0 - 1 : 0;
        out_lo = w[p0 >> 4];
        out_hi = w[(p0 >> 4) + 1];
    }

    __device__ __forceinline__ void select(bool e, u64 mh, u32 mp) {
        if (!DIRECT) {
            const bool keep = e && cnt < (u32)CAP;
            const u32 sl = (keep ? cnt : (u32)CAP) * 65u + (u32)lane;
            st.sh[sl] = mh;
            st.sp[sl] = mp;
        } else if (e) {
            ghash[gbase + cnt] = mh;
            gpos[gbase + cnt] = mp;
        }
        cnt += e ? 1u : 0u;
    }

    template <bool FIRST>
    __device__ __forceinline__ void block(u32 i0) {
        const u32 t0 = i0 + (u32)k - 1;
        const u32 cinb = __builtin_amdgcn_alignbit(in_hi, in_lo, (t0 & 15) * 2);
        u32 coutb;
        if (FIRST) coutb = out_lo << 2;  // slot 0: nothing leaves; slot o >= 1 sees base o-1
        else coutb = __builtin_amdgcn_alignbit(out_hi, out_lo, ((i0 - 1) & 15) * 2);
        // all W table fetches first: they do not depend on the hash chain
        uint4 xs[W];
#pragma unroll
        for (int o = 0; o < W; ++o) {
            const u32 cin = (cinb >> (2 * o)) & 3;
            u32 idx = (((coutb >> (2 * o)) & 3) << 2) | cin;
            if (FIRST && o == 0) idx = 16 + cin;
            xs[o] = xt[idx];
        }
        load_block_words(i0 + W);  // next block's words: in flight while this block is hashed
#pragma unroll
        for (int o = 0; o < W; ++o) {
            const u32 i = i0 + o;
            fh = hrol1(fh);
            rh = hror1(rh);
            fh.lo ^= xs[o].x;
            fh.hi ^= xs[o].y;
            rh.lo ^= xs[o].z;
            rh.hi ^= xs[o].w;
            const u64 f64 = to64(fh), r64 = to64(rh);
            const bool rev = r64 < f64;
            const u64 h = rev ? r64 : f64;
            const u32 ps = rev ? (i | 0x80000000u) : i;
            if (o == 0) {
                Ph = h;
                Pp = ps;
            } else if (h < Ph) {
                Ph = h;
                Pp = ps;
            }
            if (!FIRST || o == W - 1) {
                u64 mh = Ph;
                u32 mp = Pp;
                if (o != W - 1) {
                    const bool takeS = !(Ph < bh[o + 1]);
                    mh = takeS ? bh[o + 1] : Ph;
                    mp = takeS ? bp[o + 1] : Pp;
                }
                const bool e = (i < nk) & (mp != prev);
                prev = mp;
                select(e, mh, mp);
            }
            bh[o] = h;
            bp[o] = ps;
        }
        if (FIRST && !DIRECT) {
#pragma unroll
            for (int a = 0; a + 1 < W; ++a)
#pragma unroll
                for (int b = a + 1; b < W; ++b) tie |= (bh[a] == bh[b]) ? 1u : 0u;
        }
#pragma unroll
        for (int q = W - 2; q >= 0; --q) {
            const bool t = bh[q + 1] < bh[q];
            bh[q] = t ? bh[q + 1] : bh[q];
            bp[q] = t ? bp[q + 1] : bp[q];
        }
    }

    __device__ __forceinline__ void run(u32 nk_max) {
        fh.lo = fh.hi = rh.lo = rh.hi = 0;
        Ph = 0;
        Pp = 0;
        prev = 0xffffffffu;
        cnt = 0;
        tie = 0;
        // warm-up: bases 0..k-2 enter, nothing leaves (table rows 16..19)
        for (int t0 = 0; t0 < k - 1; t0 += 16) {
            const u32 word = w[t0 >> 4];
            const int nb = (k - 1 - t0) < 16 ? (k - 1 - t0) : 16;
            for (int j = 0; j < nb; ++j) {
                const uint4 x = xt[16 + ((word >> (2 * j)) & 3)];
                fh = hrol1(fh);
                rh = hror1(rh);
                fh.lo ^= x.x;
                fh.hi ^= x.y;
                rh.lo ^= x.z;
                rh.hi ^= x.w;
            }
        }
        load_block_words(0);
        block<true>(0);
        for (u32 i0 = W; i0 < nk_max; i0 += W) block<false>(i0);
    }
};

template <int W, int CAP>
__global__ __launch_bounds__(64) void k_minimizer_fast(KArgs a) {
    __shared__ uint4 s_tab[32];
    __shared__ u64 s_h[FStage<CAP>::SLOTS];
    __shared__ u32 s_p[FStage<CAP>::SLOTS];
    __shared__ u16 s_m[CAP * 64];
    const int lane = lane_id();
    build_xtab(s_tab, a.k, lane);
    __syncthreads();
    FStage<CAP> st{s_h, s_p, s_m};
    for (;;) {
        const u32 unit = next_ticket(a.ticket, lane);
        if (unit >= a.nunits) break;
        const u64 r = (u64)unit * 64 + lane;
        u64 off = 0, L = 0;
        if (r < a.n) {
            const u64 d = a.desc[r];
            off = d >> 24;
            L = d & 0xffffffULL;
        }
        const bool ok = r < a.n && L >= (u64)a.circ_ext && (L - (u64)a.circ_ext) + 1 >= (u64)a.k + (u64)W;
        const u32 nk = ok ? (u32)(L - a.k + 1) : 0u;
        const u32 nk_max = wave_max_u32(nk);
        u32 cnt = 0, tie = 0;
        if (nk_max) {
            FastMin<W, CAP, false> fm;
            fm.w = a.words + off;
            fm.xt = s_tab;
            fm.k = a.k;
            fm.lane = lane;
            fm.nk = nk;
            fm.st = st;
            fm.ghash = a.hash;
            fm.gpos = a.pos;
            fm.gbase = 0;
            fm.run(nk_max);
            cnt = fm.cnt;
            tie = fm.tie;
        }
        // ---- unit epilogue: scan, look-back, LDS -> HBM copy-out, CSR offsets ----
        const u32 incl = wave_incl_scan_u32(cnt, lane);
        const u32 excl = incl - cnt;
        const u32 T = wave_bcast_u32(incl, 63);
        const u64 base = (a.debug & 1) ? (u64)unit * (64 * CAP) : lookback_exclusive(a.lookback, unit, (u64)T, lane);
        const bool ovf = base + T > a.cap;
        const bool any_over = __ballot(cnt > (u32)CAP) != 0;
        if (a.debug & 2) {
        } else if (!ovf && !any_over) {
            const u32 cmax = wave_max_u32(cnt);
            for (u32 e = 0; e < cmax; ++e)
                if (e < cnt) st.smap[excl + e] = (u16)(e * 65u + (u32)lane);
            __syncthreads();
            for (u32 t = lane; t < T; t += 64) {
                const u32 sl = st.smap[t];
                a.hash[base + t] = st.sh[sl];
                a.pos[base + t] = st.sp[sl];
            }
            __syncthreads();
        } else if (!ovf) {  // rare: some lane selected more than CAP tuples -> recompute, store straight to HBM
            FastMin<W, CAP, true> fm;
            fm.w = a.words + off;
            fm.xt = s_tab;
            fm.k = a.k;
            fm.lane = lane;
            fm.nk = nk;
            fm.st = st;
            fm.ghash = a.hash;
            fm.gpos = a.pos;
            fm.gbase = base + excl;
            fm.run(nk_max);
        } else if (lane == 0) {
            atomicOr(&a.ticket[1], 1u);
        }
        if (r < a.n) {
            a.offsets[r + 1] = base + incl;
            u8 sbyte = ok ? BSK_ST_OK : BSK_ST_SHORT;
            if (tie) sbyte |= BSK_ST_FIRST_WINDOW_TIE;
            if (ok && a.rflags) sbyte |= a.rflags[r];
            a.status[r] = sbyte;
        }
        if (unit == 0 && lane == 0) a.offsets[0] = 0;
        if (unit == a.nunits - 1 && lane == 63) *a.total = base + incl;
    }
}

// ---- dispatch table --------------------------------------------------------------------
#define BSK_FAST_CAP 32
#define BSK_FAST_WS(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)

static inline bool fast_minimizer_supported(int w) { return w >= 2 && w <= 16; }

static inline int fast_minimizer_blocks_per_cu(int w) {
    int nb = 0;
    hipError_t e = hipErrorInvalidValue;
    switch (w) {
#define X(WW) \
    case WW: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_minimizer_fast<WW, BSK_FAST_CAP>, 64, 0); break;
        BSK_FAST_WS(X)
#undef X
        default: break;
    }
    if (e != hipSuccess || nb < 1) {
        (void)hipGetLastError();
        nb = 1;
    }
    return nb;
}

static inline void fast_minimizer_launch(int w, int grid, hipStream_t stream, const KArgs &a) {
    switch (w) {
#define X(WW) \
    case WW: hipLaunchKernelGGL((k_minimizer_fast<WW, BSK_FAST_CAP>), dim3(grid), dim3(64), 0, stream, a); break;
        BSK_FAST_WS(X)
#undef X
        default: break;
    }
}

}  // namespace bsk
