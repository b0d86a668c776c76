// kernels_seg.hpp -- k_minimizer_seg<W>: the minimizer kernel at three wavefronts per SIMD (12 per CU instead of 8).
//
// What limits k_minimizer_fast is not HBM and not the VALU but the instructions each wavefront can issue (one per ~5 cycles)
// and the time it sits in LDS / copy-out latencies; throughput is linear in the resident wavefronts.  A third
// wavefront per SIMD needs <= 168 VGPRs and <= 13.6 KB of LDS, and the staging of a whole unit's tuples (64 reads x ~22 tuples
// x 10 B = 14 KB before any margin) cannot fit that.  So this kernel does not stage a whole read:
//   * every read owns a fixed slab of `slab_read` tuples in the result arrays (as k_minimizer_dense / the protein kernel do),
//     so the destination of a tuple is known as soon as it is selected -- no unit-wide prefix is needed to place it;
//   * the wavefront empties its staging every NB blocks ("segment"): ALL staged tuples leave, whatever their alignment.  A
//     hash line of a slab is therefore written in two or three pieces by consecutive flushes; they are plain (not
//     non-temporal) stores, so the pieces meet in the L2 (measured on k_minimizer_dense: half-line pieces cost +4 % write
//     bytes as plain stores);
//   * the staging is the paired-column layout of k_minimizer_fast with R = NB*W + 1 rows per pair of lanes: one lane alone
//     cannot fill a column within a segment (it stages at most one tuple per step), so the k-mer loop needs no bounded-store
//     variant at all; the two lanes of a pair meeting in the middle is detected after the fact (count sum >= R) and sends the
//     unit to the recompute-and-store-directly path, like a read that outgrows its slab;
//   * the table rows of a block are fetched in chunks of 4 (not all W up front): 144 VGPRs.
//
// MEASURED (10^8 x 150 bp, k=21 w=11, 12 waves per CU, same digest): 25.5 ms = 589 Gbases/s against 17.2 ms for k_minimizer_fast
// at 8 waves.  HBM traffic 15.8 GB fetched + 47.2 GB written against 4.6 + 27.3 GB algorithmic: every flush touches 2-3
// lines per read with ~6 tuples each, the per-XCD L2 (4 MB) cannot keep the 9 MB of half-written slab lines of its resident
// reads until the next flush, and a partial line costs the memory system as much as a whole one (scripts/ubench/
// partial_write.hip: ~21 G line writes per second whatever the piece size); with plain stores the outputs also push the
// reads' own lines out of the L2 (input fetched 3.4x).  The extra wavefronts are worth less than that: the kernel stays
// an experiment (BSK_SEG=1) and k_minimizer_fast (whole unit staged, one contiguous copy-out, 8 waves) stays the plan.
#pragma once
#include "kernels_fast.hpp"
#include "kernels_wpr.hpp"

namespace bsk {

template <int W>
struct SegCfg {
    static constexpr int NB = (35 / W) > 0 ? (35 / W) : 1;  // blocks per segment
    static constexpr int R = NB * W + 1;                    // rows of a paired column
    typedef PLds<R, true> LY;
    static constexpr int DST = LY::TOTAL;                   // u32 [64]: slab position of a lane's first staged tuple - its exclusive offset
    static constexpr int TOTAL = DST + 256;
};

// segment copy-out: the staged tuples of all lanes, in read order, to the per-read slabs (plain stores: partial lines)
template <int W, int U>
__device__ __forceinline__ void seg_copyout(char *lds, int lane, u32 cnt, u32 excl, u32 T, u32 dst0, u64 ubase, const KArgs &a) {
    typedef typename SegCfg<W>::LY LY;
    constexpr int R = SegCfg<W>::R;
    u32 *s_excl = reinterpret_cast<u32 *>(lds + LY::EXCL);
    u32 *s_dst = reinterpret_cast<u32 *>(lds + SegCfg<W>::DST);
    u64 *s_heads = reinterpret_cast<u64 *>(lds + LY::HEADS);
    u8 *s_nz = reinterpret_cast<u8 *>(lds + LY::NZ);
    const u64 nzmask = __builtin_amdgcn_ballot_w64(cnt > 0);
    s_excl[lane] = excl;
    s_dst[lane] = dst0 - excl;  // destination (relative to the unit's slabs) of output t of this lane = s_dst + t
    if (lane < LY::NHEADS) s_heads[lane] = 0;
    wave_sync_lds();
    if (cnt > 0) {
        s_nz[__builtin_amdgcn_mbcnt_hi((u32)(nzmask >> 32), __builtin_amdgcn_mbcnt_lo((u32)nzmask, 0))] = (u8)lane;
        atomicOr(&s_heads[excl >> 6], 1ULL << (excl & 63));
    }
    wave_sync_lds();
    u32 heads_before = 0;
    const u64 *sh = reinterpret_cast<const u64 *>(lds + LY::SH);
    for (u32 t0 = 0; t0 < T; t0 += 64 * U) {
        u32 rank[U], owner[U], ex[U], dd[U], sl[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const u32 c = (t0 >> 6) + j;
            const u64 M = s_heads[c < (u32)(LY::NHEADS - 1) ? c : (u32)(LY::NHEADS - 1)];  // the last word is never set
            const u32 below = __builtin_amdgcn_mbcnt_hi((u32)(M >> 32), __builtin_amdgcn_mbcnt_lo((u32)M, 0));
            rank[j] = heads_before + below + (u32)((M >> lane) & 1) - 1;
            heads_before += (u32)__builtin_popcountll(M);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) owner[j] = s_nz[rank[j]];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            ex[j] = s_excl[owner[j]];
            dd[j] = s_dst[owner[j]];
        }
        u64 hv[U];
        u32 pv[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const u32 t = t0 + 64 * j + lane;
            const u32 e = t - ex[j];
            sl[j] = t < T ? (owner[j] < 32u ? e : (u32)(R - 1) - e) * LY::ROW + (owner[j] & 31u) : 0u;
            hv[j] = sh[sl[j]];
            pv[j] = *reinterpret_cast<const u16 *>(lds + LY::SP + sl[j] * 2);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const u32 t = t0 + 64 * j + lane;
            if (t < T) {
                const u64 d = ubase + (u64)(dd[j] + t);
#ifndef BSK_SEG_NOSTORE  // (dev: timing of the 12-wave compute alone; results invalid)
                a.hash[d] = hv[j];
                a.pos[d] = (pv[j] & 0x7fffu) | ((pv[j] & 0x8000u) << 16);
#else
                if (hv[j] == 0x123456789abcdefULL) a.pos[d] = pv[j];
#endif
            }
        }
    }
    wave_sync_lds();
}

template <int W>
__global__ __launch_bounds__(64, 3) void k_minimizer_seg(KArgs a) {
    constexpr int NB = SegCfg<W>::NB, R = SegCfg<W>::R;
    typedef typename SegCfg<W>::LY LY;
    typedef FastMin<W, BSK_FAST_CAP, true, false, true, false, R, 4> FM;
    __shared__ __attribute__((aligned(16))) char lds[SegCfg<W>::TOTAL];
    LDSQ char *const ldsq = (LDSQ char *)lds;
    const int lane = lane_id();
    build_xtab(reinterpret_cast<uint4 *>(lds + LY::TAB), a.k, lane);
    __syncthreads();
    const u32 slab_read = (u32)a.slab_read;
    const u32 col8 = (u32)(lane & 31) * 8u;
    const bool up = lane < 32;
    const u32 slim = (u32)((R - 1) * LY::ROW * 8) + col8;
    const u32 slot0 = up ? col8 : slim;
    u64 d_next = 0;
    u32x4 pw_next = {0, 0, 0, 0};
    bool pre = false;
    for (u32 unit = next_ticket(a.ticket, lane) * 8u, uend = unit + 8u; unit < a.nunits; ++unit, ({
             if (unit == uend) {
                 unit = next_ticket(a.ticket, lane) * 8u;
                 uend = unit + 8u;
             }
         })) {
        const u64 r = (u64)unit * 64 + lane;
        u64 d;
        u32x4 pw;
        if (pre) {  // descriptor and first words were requested one unit ahead (k_minimizer_fast)
            d = d_next;
            pw = pw_next;
        } else {
            d = r < a.n ? a.desc[r] : 0;
            pw = *reinterpret_cast<const GLBQ u32x4_u *>((size_t)(a.words + (d >> 24)));
        }
        const u64 off = d >> 24, L = d & 0xffffffULL;
        const bool nxt = unit + 1 != uend && unit + 1 < a.nunits;
        if (nxt) d_next = r + 64 < a.n ? a.desc[r + 64] : 0;
        const bool ok = r < a.n && L >= (u64)a.circ_ext && (L - (u64)a.circ_ext) + 1 >= (u64)a.k + (u64)W;
        const u32 nk = ok ? (u32)(L - a.k + 1) : 0u;
        const u32 nk_max = wave_max_u32(nk);
        const u64 ubase = (u64)unit * 64 * slab_read;
        u32 done = 0, tie = 0;
        bool over = false;  // wave-uniform: some lane outgrew its slab, or a pair of lanes its column
        if (nk_max) {
            FM fm;
            fm.w = a.words + off;
            fm.pw = pw;
            fm.lds = ldsq;
            fm.k = a.k;
            fm.lane = lane;
            fm.nk = nk;
            fm.begin();
            fm.template block<true, false>(0);  // one first, one steady variant: see FastMin::run
            int inseg = 1;
            for (u32 i0 = W;; i0 += W) {
                const bool fin = i0 >= nk_max;
                if (inseg == NB || fin) {  // end of a segment: everything staged leaves
                    inseg = 0;
                    const u32 cnt = (up ? fm.slot : slim + col8 - fm.slot) / (u32)(LY::ROW * 8);  // col8 < ROW*8: the quotient is the row count
                    const u32 cnt_pair = cnt + (u32)__builtin_amdgcn_ds_bpermute((lane ^ 32) * 4, (int)cnt);
                    over = over || __builtin_amdgcn_ballot_w64(cnt_pair >= (u32)R || done + cnt > slab_read) != 0;
                    if (!over) {
                        const u32 incl = wave_incl_scan_u32(cnt, lane);
                        const u32 T = wave_bcast_u32(incl, 63);
                        if (T) seg_copyout<W, 2>(lds, lane, cnt, incl - cnt, T, (u32)lane * slab_read + done, ubase, a);
                    }
                    done += cnt;
                    fm.slot = slot0;
                }
                if (fin) break;
                fm.template block<false, false>(i0);
                ++inseg;
            }
            tie = fm.tie;
        }
        if (nxt) pw_next = *reinterpret_cast<const GLBQ u32x4_u *>((size_t)(a.words + (d_next >> 24)));
        pre = nxt;
        u64 first = ubase + (u64)lane * slab_read;
        if (over) {
            // rare: recompute the unit and store straight to the overflow region (done = the exact tuple count of every lane)
            const u32 incl = wave_incl_scan_u32(done, lane);
            const u32 T = wave_bcast_u32(incl, 63);
            u64 ob = 0;
            if (lane == 0) ob = atomicAdd(a.total + 1, (u64)T);
            ob = wave_bcast_u64(ob, 0);
            if (ob + T <= a.ovf_cap) {
                first = a.ovf_base + ob + (incl - done);
                FastMin<W, BSK_FAST_CAP, true, true> fd;
                fd.w = a.words + off;
                fd.pw = pw;
                fd.lds = ldsq;
                fd.k = a.k;
                fd.lane = lane;
                fd.nk = nk;
                fd.ghash = a.hash;
                fd.gpos = a.pos;
                fd.gbase = first;
                fd.run(nk_max);
            } else {
                done = 0;  // result buffers too small: flagged, the host re-runs with a larger overflow region
                if (lane == 0) atomicOr(&a.ticket[1], 1u);
            }
        }
        if (r < a.n) {
            a.refs[r] = (first << 24) | done;
            u8 sbyte = ok ? BSK_ST_OK : BSK_ST_SHORT;
            if (tie && ok) sbyte |= BSK_ST_FIRST_WINDOW_TIE;
            if (ok && a.rflags) sbyte |= a.rflags[r];
            a.status[r] = sbyte;
        }
    }
}

#ifdef BSK_IMPL_SEG  // dispatch functions: compiled in the family's own translation unit (k_minimizer_seg.hip)
#ifndef BSK_SEG_WS
#define BSK_SEG_WS(X) X(5) X(11) X(15)  // an experiment: never planned unless BSK_SEG=1
#endif
bool seg_minimizer_supported(int w) {
    switch (w) {
#define X(WW) case WW:
        BSK_SEG_WS(X)
#undef X
        return true;
        default: return false;
    }
}
int seg_minimizer_blocks_per_cu(int w) {
    int nb = 0;
    hipError_t e = hipErrorInvalidValue;
    switch (w) {
#define X(WW) \
    case WW: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_minimizer_seg<WW>, 64, 0); break;
        BSK_SEG_WS(X)
#undef X
        default: break;
    }
    if (e != hipSuccess || nb < 1) {
        (void)hipGetLastError();
        nb = 1;
    }
    return nb;
}
void seg_minimizer_launch(int w, int grid, hipStream_t stream, const KArgs &a) {
    switch (w) {
#define X(WW) \
    case WW: hipLaunchKernelGGL((k_minimizer_seg<WW>), dim3(grid), dim3(64), 0, stream, a); break;
        BSK_SEG_WS(X)
#undef X
        default: break;
    }
}
void wpr_minimizer_launch(int grid, hipStream_t stream, const KArgs &a) {
    hipLaunchKernelGGL((k_minimizer_wpr<11>), dim3(grid), dim3(64), 0, stream, a);
}
int wpr_minimizer_blocks_per_cu() {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_minimizer_wpr<11>, 64, 0) != hipSuccess || nb < 1) {
        (void)hipGetLastError();
        nb = 1;
    }
    return nb;
}
#endif  // BSK_IMPL_SEG

}  // namespace bsk
