// kernels_simhash.hpp -- SimHash iterator (NewSimHashIterator / NextSimHash, iterator.go:113-612) with bit-sliced
// counters, 2-bit input, one read per lane.
//
// Per k-mer i the reference keeps 64 int16 counters: sum[b] = how many of the k-m+1 FracMinHash-filtered m-mer hashes
// inside the k-mer have bit b set (iterator.go:291-354), nPos = how many hashes are present, and emits bit b = 1 iff
// sum[b] >= (nPos+1)/2 (:360-428).  Moving from k-mer i-1 to i removes m-mer i-1 and adds m-mer i+nh-1, nh = k-m+1
// (:434-523).  Here the 64 counters of a lane are PL bit planes of 64 bits (counter of hash bit j = bit j of the planes):
//   * add/remove one hash = a ripple carry / borrow through the planes (3 boolean ops per 32-bit half plane),
//   * "sum >= t" for all 64 counters at once = the borrow out of sum - t, one 3-input boolean op (v_bitop3) per half plane,
// about 110 VALU ops per k-mer instead of ~400 for 64 scalar counters.  The leaving m-mer hash is not kept in a ring
// (63 x 512 B of LDS per wavefront) but recomputed by a second rolling hasher that lags nh positions behind.
// Output: the 16-value tile / line-padded runs of k_nthash_fast.
#pragma once
#include "kernels_fast.hpp"

namespace bsk {

// 16 consecutive 2-bit codes starting at base position pos of the lane's read (staged words sw[word*64 + lane]);
// pos may be -1 or -2 (first block): the missing leading codes read as 0
__device__ __forceinline__ u32 codes16(LDSQ const u32 *sw, int lane, int pos) {
    if (pos < 0) return sw[lane] << (2 * (-pos));  // wave-uniform branch
    const u32 wi = (u32)pos >> 4;
    return __builtin_amdgcn_alignbit(sw[(wi + 1) * 64 + lane], sw[wi * 64 + lane], ((u32)pos & 15) * 2);
}

struct RollNt {  // ntHash-1 of the current m-mer, both strands
    u32 fl, fh, rl, rh;
    __device__ __forceinline__ void roll(u32x4 x) {
        const u32 p = __builtin_amdgcn_alignbit(fl, fh, 31), q = __builtin_amdgcn_alignbit(fh, fl, 31);
        const u32 c = __builtin_amdgcn_alignbit(rh, rl, 1), d = __builtin_amdgcn_alignbit(rl, rh, 1);
        fl = p ^ x.x;
        fh = q ^ x.y;
        rl = c ^ x.z;
        rh = d ^ x.w;
    }
};

// NWL = packed words of a read staged in LDS: 12 (reads of up to 160 bases) leaves 12 waves per CU, 20 (288 bases) 10, 34 (512 bases) 8
#define BSK_SIM_SHORT_WORDS 12
#define BSK_SIM_MID_WORDS 20
template <int PL, int NWL = BSK_NT_FAST_WORDS>
__global__ __launch_bounds__(64) void k_simhash_fast(KArgs a) {
    constexpr int TL = 18;
    constexpr int SW_OFF = 512 + 64 * TL * 8;
    __shared__ __attribute__((aligned(16))) char lds[SW_OFF + NWL * 64 * 4];
    __shared__ u64 s_off[64];
    __shared__ u32 s_nk[64];
    __shared__ u64 s_base[8];
    LDSQ char *const lq = (LDSQ char *)lds;
    LDSQ u32 *const sw = reinterpret_cast<LDSQ u32 *>(lq + SW_OFF);
    const int lane = lane_id();
    build_xtab(reinterpret_cast<uint4 *>(lds), a.m, lane);
    __syncthreads();
    const int k = a.k, m = a.m, nh = a.k - a.m + 1;
    const bool canon = a.canonical != 0;
    const u64 maxhash = a.scale > 1 ? 0xffffffffffffffffULL / (u64)a.scale : 0xffffffffffffffffULL;  // iterator.go:181-185
    const u32 mh_lo = (u32)maxhash, mh_hi = (u32)(maxhash >> 32);
    for (;;) {
      const u32 u0 = next_ticket(a.ticket, lane) * 8u;
      if (u0 >= a.nunits) break;
      const u32 u1 = u0 + 8u < a.nunits ? u0 + 8u : a.nunits;
      if (!a.uniform_len) {  // ragged batch: one look-back per ticket, resolved before its units are processed (see k_nthash_fast)
          u64 run = 0;
          for (u32 unit = u0; unit < u1; ++unit) {
              const u64 r = (u64)unit * 64 + lane;
              u64 L = 0;
              if (r < a.n) L = a.desc[r] & 0xffffffULL;
              const bool ok = r < a.n && L >= (u64)a.circ_ext && (L - (u64)a.circ_ext) >= (u64)k;
              const u32 pk = ok ? ((u32)(L - k + 1) + 15u) & ~15u : 0u;
              if (lane == 0) s_base[unit - u0] = run;
              run += wave_sum_u64((u64)pk);
          }
          const u64 tbase = lookback_exclusive(a.lookback, u0 >> 3, run, lane);
          wave_sync_lds();
          if (lane < 8) s_base[lane] += tbase;
          wave_sync_lds();
      }
      for (u32 unit = u0; unit < u1; ++unit) {
        const u64 r = (u64)unit * 64 + lane;
        u64 off = 0, L = 0;
        if (r < a.n) {
            const u64 d = a.desc[r];
            off = d >> 24;
            L = d & 0xffffffULL;
        }
        const bool ok = r < a.n && L >= (u64)a.circ_ext && (L - (u64)a.circ_ext) >= (u64)k;  // iterator.go:128
        const u32 nk = ok ? (u32)(L - k + 1) : 0u;
        const u32 nk_max = wave_max_u32(nk);
        const u32 pk = (nk + 15u) & ~15u;
        const u64 incl = wave_incl_scan_u64((u64)pk, lane);
        const u64 T = wave_bcast_u64(incl, 63);
        const u64 base = a.uniform_len ? (u64)unit * 64 * ((nk_max + 15u) & ~15u) : s_base[unit - u0];
        const bool ovf = base + T > a.cap;
        if (ovf && lane == 0) atomicOr(&a.ticket[1], 1u);
        if (r < a.n) {
            a.refs[r] = ((base + incl - pk) << 24) | nk;
            u8 sbyte = ok ? BSK_ST_OK : BSK_ST_SHORT;
            if (ok && a.rflags) sbyte |= a.rflags[r];
            a.status[r] = sbyte;
        }
        if (unit == a.nunits - 1 && lane == 63) *a.total = base + incl;
        if (ovf || nk_max == 0) continue;
        s_off[lane] = base + incl - pk;
        s_nk[lane] = pk;
        wave_sync_lds();
        u64 roff[8];
        u32 rnk[8];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            roff[rr] = s_off[rr * 8 + (lane >> 3)] + (u32)(lane & 7) * 2;
            rnk[rr] = s_nk[rr * 8 + (lane >> 3)];
        }
        const u32 *__restrict__ w = a.words + off;
        const u32 nw_max = ((nk_max + (u32)k - 1 + 15) >> 4) + 2;  // <= NWL (checked by the host)
        for (u32 j = 0; j < nw_max; ++j) sw[j * 64 + lane] = w[j];
        wave_sync_lds();

        u32 pl_lo[PL], pl_hi[PL];  // bit planes of the 64 counters
#pragma unroll
        for (int p = 0; p < PL; ++p) pl_lo[p] = pl_hi[p] = 0;
        u32 npos = 0;
        RollNt A{0, 0, 0, 0}, B{0, 0, 0, 0};
        auto present = [&](const RollNt &h, u32 &xl, u32 &xh) {  // canonical / forward hash, FracMinHash filter (:279-285)
            u32 hl = h.fl, hh = h.fh;
            if (canon) {
                const lmask rev = lt64(h.rl, h.rh, h.fl, h.fh);
                hl = sel(rev, h.rl, h.fl);
                hh = sel(rev, h.rh, h.fh);
            }
            const lmask big = lt64(mh_lo, mh_hi, hl, hh);  // hash > MaxUint64/scale: treated as absent
            xl = sel(big, 0u, hl);
            xh = sel(big, 0u, hh);
        };
        auto update = [&](u32 nl, u32 nh_, u32 ol, u32 oh) {  // counters += bits(new) - bits(old)
            u32 cxl = nl & ~ol, cxh = nh_ & ~oh;  // increment where new has the bit and old has not
            u32 cyl = ol & ~nl, cyh = oh & ~nh_;  // decrement in the opposite case
#pragma unroll
            for (int p = 0; p < PL; ++p) {
                const u32 l = pl_lo[p], h = pl_hi[p];
                pl_lo[p] = l ^ (cxl | cyl);
                pl_hi[p] = h ^ (cxh | cyh);
                cxl &= l;
                cxh &= h;
                cyl &= ~l;
                cyh &= ~h;
            }
            const u32 tn = nl | nh_, to = ol | oh;  // a hash of 0 counts as absent (:288-290); (t | -t) >> 31 = (t != 0), no VCC
            npos += ((tn | (0u - tn)) >> 31) - ((to | (0u - to)) >> 31);
        };
        // warm-up: both hashers take the first m-1 bases; A then adds m-mers 0 .. nh-2 (no code is emitted yet)
        for (int t = 0; t < m - 1; ++t) {
            const u32 b = (sw[(t >> 4) * 64 + lane] >> (2 * (t & 15))) & 3;
            const u32x4 x = *reinterpret_cast<LDSQ const u32x4 *>(lq + 256 + (b << 4));
            A.roll(x);
            B.roll(x);
        }
        for (int t = 0; t < nh - 1; ++t) {
            const int pin = t + m - 1, pout = t - 1;
            const u32 bi = (sw[(pin >> 4) * 64 + lane] >> (2 * (pin & 15))) & 3;
            u32 idx = 0x100u | (bi << 4);
            if (pout >= 0) idx = (((sw[(pout >> 4) * 64 + lane] >> (2 * (pout & 15))) & 3) << 6) | (bi << 4);
            A.roll(*reinterpret_cast<LDSQ const u32x4 *>(lq + idx));
            u32 xl, xh;
            present(A, xl, xh);
            update(xl, xh, 0u, 0u);
        }
        LDSQ char *const myrow = lq + 512 + lane * (TL * 8);
        for (u32 i0 = 0; i0 < nk_max; i0 += 16) {
            // k-mer i = i0+o: A takes base i+k-1 and drops base i+nh-2; B (m-mer i-1, removed) takes base i+m-2 and drops i-2
            const u32 ain = codes16(sw, lane, (int)i0 + k - 1), aout = codes16(sw, lane, (int)i0 + nh - 2);
            const u32 bin = codes16(sw, lane, (int)i0 + m - 2), bout = codes16(sw, lane, (int)i0 - 2);
#pragma unroll
            for (int o = 0; o < 16; ++o) {
                u32 ia = ((o >= 2 ? (ain >> (2 * o - 4)) : (ain << (4 - 2 * o))) & 0x30u) |
                         ((o >= 3 ? (aout >> (2 * o - 6)) : (aout << (6 - 2 * o))) & 0xC0u);
                if (o == 0 && i0 == 0 && nh == 1) ia = (ia & 0x30u) | 0x100u;  // k == m: the first m-mer has no predecessor
                A.roll(*reinterpret_cast<LDSQ const u32x4 *>(lq + ia));
                u32 nl, nhh, ol = 0, oh = 0;
                present(A, nl, nhh);
                if (o > 0 || i0 > 0) {  // k-mer 0 removes nothing
                    u32 ib = ((o >= 2 ? (bin >> (2 * o - 4)) : (bin << (4 - 2 * o))) & 0x30u) |
                             ((o >= 3 ? (bout >> (2 * o - 6)) : (bout << (6 - 2 * o))) & 0xC0u);
                    if (o == 1 && i0 == 0) ib = (ib & 0x30u) | 0x100u;  // m-mer 0 has no predecessor
                    B.roll(*reinterpret_cast<LDSQ const u32x4 *>(lq + ib));
                    present(B, ol, oh);
                }
                update(nl, nhh, ol, oh);
                // code bit j = 1 iff counter_j >= t, t = (nPos+1)/2 (:360): no borrow out of counter_j - t
                const u32 thr = (npos + 1u) >> 1;
                u32 bl = 0, bh = 0;
#pragma unroll
                for (int p = 0; p < PL; ++p) {
                    const u32 tb = (u32)__builtin_amdgcn_sbfe((int)thr, p, 1);  // bit p of t, replicated
                    bl = (~pl_lo[p] & tb) | (~(pl_lo[p] ^ tb) & bl);
                    bh = (~pl_hi[p] & tb) | (~(pl_hi[p] ^ tb) & bh);
                }
                const u32 some = 0u - ((npos | (0u - npos)) >> 31);  // nPos == 0: the code is 0 (:356-358)
                *reinterpret_cast<LDSQ u64 *>(myrow + o * 8) = ((u64)(~bh & some) << 32) | (~bl & some);
            }
            wave_sync_lds();
            u32x4 tv[8];
#pragma unroll
            for (int rr = 0; rr < 8; ++rr)
                tv[rr] = *reinterpret_cast<LDSQ const u32x4 *>(lq + 512 + (rr * 8 + (lane >> 3)) * (TL * 8) + (lane & 7) * 16);
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                if (i0 + (u32)(lane & 7) * 2 < rnk[rr]) {
                    u64x2_a8 vv;
                    vv.a = ((u64)tv[rr].y << 32) | tv[rr].x;
                    vv.b = ((u64)tv[rr].w << 32) | tv[rr].z;
                    nt_store_u64x2(a.hash + roff[rr] + i0, vv.a, vv.b);
                }
            }
            wave_sync_lds();
        }
      }
    }
}

}  // namespace bsk
