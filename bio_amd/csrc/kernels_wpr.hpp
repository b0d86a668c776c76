// kernels_wpr.hpp -- k_minimizer_wpr<W>: the mapping the task statement sketches, built to be MEASURED against k_minimizer_fast
// (DESIGN.md 3.1 "A/B"): ONE READ PER WAVEFRONT, lanes = positions.
//
//   * the wave walks its read in chunks of 64 bases; lane l owns base t = 64c + l and produces the canonical hash of the k-mer
//     that ENDS there, without any rolling state:  with  A_t = ror(seed[s_t], t),  B_t = rol(seedc[s_t], t)  and the running
//     XORs  QA_t = A_0 ^ ... ^ A_t,  QB_t likewise,
//         fwd(i) = rol(QA_e ^ QA_{e-k}, e),   rev(i) = ror(QB_e ^ QB_{e-k}, i),      e = i + k - 1  (the k-mer's last base),
//     every rotation amount being a function of the LANE only (t mod 64 = l): two v_alignbit + two selects each;
//   * QA / QB are wave-wide inclusive XOR scans: DPP row_shr 1,2,4,8 + row_bcast 15/31 (6 steps x 4 registers) plus the carry
//     of the previous chunks (v_readlane of lane 63);
//   * Q_{e-k} lives k lanes below, or in the previous chunk: one select (previous / current chunk) + one ds_bpermute per register;
//   * window of W k-mers ending at lane l: log-step leftmost minima with look-back 1, 2, 4 and a final look-back 3
//     (8 + 4 overlapping = 11 for W = 11), each a select + ds_bpermute of (hash, pos) and a 64-bit compare;
//   * a window's minimum is emitted when its position differs from the previous lane's: ballot + mbcnt give the tuple's slot,
//     the selected lanes store (hash, pos|strand) straight to the read's slab -- no LDS transposition, no staging at all.
// LDS: a 4-entry seed table.  Registers: ~60.  Everything the other mapping pays for staging and copy-out is gone; what it
// costs instead is ~130 wave-instructions per 64 POSITIONS of one read, against ~43 per k-mer step of 64 READS.
#pragma once
#include "kernels_fast.hpp"

namespace bsk {

template <int CTRL, int ROWMASK>
__device__ __forceinline__ u32 dpp_xor(u32 v) {  // v ^= v[source lane of CTRL]; lanes without a source, or outside ROWMASK, keep v
    return v ^ (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, false);
}
__device__ __forceinline__ u32 wave_xor_scan(u32 v) {  // inclusive, lanes 0..63
    v = dpp_xor<0x111, 0xf>(v);  // row_shr:1
    v = dpp_xor<0x112, 0xf>(v);  // row_shr:2
    v = dpp_xor<0x114, 0xf>(v);  // row_shr:4
    v = dpp_xor<0x118, 0xf>(v);  // row_shr:8
    v = dpp_xor<0x142, 0xa>(v);  // row_bcast:15 -> rows 1 and 3
    v = dpp_xor<0x143, 0xc>(v);  // row_bcast:31 -> rows 2 and 3
    return v;
}
// rotate right by a per-lane amount r = 32*swap + r5 (both fixed per lane for the whole kernel)
__device__ __forceinline__ void ror64v(u32 &lo, u32 &hi, u32 r5, lmask swap) {
    const u32 a = sel(swap, lo, hi), b = sel(swap, hi, lo);  // (a:b) = the value rotated by 32 when swap
    lo = __builtin_amdgcn_alignbit(a, b, r5);
    hi = __builtin_amdgcn_alignbit(b, a, r5);
}
// value of lane (l - D) of the 128-lane sequence [previous chunk | current chunk]
template <int D>
__device__ __forceinline__ u32 look_back(u32 prev, u32 cur, lmask from_prev /* lanes >= 64 - D */, u32 addr /* ((l - D) & 63) * 4 */) {
    return (u32)__builtin_amdgcn_ds_bpermute((int)addr, (int)sel(from_prev, prev, cur));
}
__device__ __forceinline__ u32 look_back_k(u32 prev, u32 cur, lmask from_prev, u32 addr) {
    return (u32)__builtin_amdgcn_ds_bpermute((int)addr, (int)sel(from_prev, prev, cur));
}

struct HP {  // hash + (k-mer index | strand << 15)
    u32 lo, hi, p;
};
// leftmost minimum of (left = the look-back value, right = own): the right one only when strictly smaller
__device__ __forceinline__ HP lmin(HP left, HP right) {
    const lmask r = lt64(right.lo, right.hi, left.lo, left.hi);
    HP o;
    o.lo = sel(r, right.lo, left.lo);
    o.hi = sel(r, right.hi, left.hi);
    o.p = sel(r, right.p, left.p);
    return o;
}
template <int D>
__device__ __forceinline__ HP look_back3(const HP &prev, const HP &cur, lmask from_prev, u32 addr) {
    HP o;
    o.lo = look_back<D>(prev.lo, cur.lo, from_prev, addr);
    o.hi = look_back<D>(prev.hi, cur.hi, from_prev, addr);
    o.p = look_back<D>(prev.p, cur.p, from_prev, addr);
    return o;
}

template <int W>
__global__ __launch_bounds__(64) void k_minimizer_wpr(KArgs a) {
    static_assert(W == 11, "the look-back ladder below is the one of W = 11 (8 + 4 overlapping)");
    __shared__ __attribute__((aligned(16))) u32 s_seed[16];  // [code] = {fwd lo, fwd hi, complement lo, complement hi}
    const int lane = lane_id();
    if (lane < 4) {
        const u64 f = seed_fwd_code((unsigned)lane), r = seed_rev_code((unsigned)lane);
        s_seed[lane * 4 + 0] = (u32)f;
        s_seed[lane * 4 + 1] = (u32)(f >> 32);
        s_seed[lane * 4 + 2] = (u32)r;
        s_seed[lane * 4 + 3] = (u32)(r >> 32);
    }
    __syncthreads();
    const int k = a.k;
    const u32 slab_read = (u32)a.slab_read;
    // per-lane rotation amounts (t mod 64 = lane): A_t = ror(seed, l); B_t = rol(seedc, l) = ror(., 64 - l); fwd = rol(., l); rev = ror(., l - k + 1)
    const u32 rA = (u32)lane, rB = (u32)(64 - lane) & 63u, rR = (u32)(lane - k + 1) & 63u;
    const lmask swA = __builtin_amdgcn_ballot_w64((rA & 32u) != 0), swB = __builtin_amdgcn_ballot_w64((rB & 32u) != 0),
                swR = __builtin_amdgcn_ballot_w64((rR & 32u) != 0);
    const u32 rA5 = rA & 31u, rB5 = rB & 31u, rR5 = rR & 31u;
    // look-back addresses and "comes from the previous chunk" masks
    const u32 adK = (u32)((lane - k) & 63) * 4u, ad1 = (u32)((lane - 1) & 63) * 4u, ad2 = (u32)((lane - 2) & 63) * 4u,
              ad3 = (u32)((lane - 3) & 63) * 4u, ad4 = (u32)((lane - 4) & 63) * 4u;
    const lmask pvK = __builtin_amdgcn_ballot_w64(lane >= 64 - k), pv1 = __builtin_amdgcn_ballot_w64(lane >= 63),
                pv2 = __builtin_amdgcn_ballot_w64(lane >= 62), pv3 = __builtin_amdgcn_ballot_w64(lane >= 61),
                pv4 = __builtin_amdgcn_ballot_w64(lane >= 60);
    for (;;) {
        const u32 tk = next_ticket(a.ticket, lane);  // 64 reads per ticket, one after the other
        if ((u64)tk * 64 >= a.n) break;
        const u64 r0 = (u64)tk * 64;
        const u64 dmine = r0 + lane < a.n ? a.desc[r0 + lane] : 0;
        const u32 nr = (u32)((a.n - r0) < 64 ? (a.n - r0) : 64);
        u32 my_cnt = 0, my_flags = 0;  // lane j collects read j's results
        u64 my_first = 0;
        for (u32 j = 0; j < nr; ++j) {
            const u64 d = wave_bcast_u64(dmine, (int)j);
            const u64 off = d >> 24;
            const u32 L = (u32)(d & 0xffffffULL);
            const bool ok = L >= (u32)a.circ_ext && (u64)(L - (u32)a.circ_ext) + 1 >= (u64)k + (u64)W;
            const u32 nk = ok ? L - (u32)k + 1 : 0u;
            const u32 *__restrict__ w = a.words + off;
            u64 base = (r0 + j) * (u64)slab_read;
            u32 limit = slab_read;
            u32 cnt = 0, tie = 0;
          for (int pass = 0; pass < 2; ++pass) {  // pass 1 (rare): the read selected more than its slab holds -- again, into the overflow region
            cnt = 0;
            u32 cAl = 0, cAh = 0, cBl = 0, cBh = 0;              // carries of the two scans
            u32 pQAl = 0, pQAh = 0, pQBl = 0, pQBh = 0;          // previous chunk's scans
            HP ph = {0, 0, 0}, pm1 = {0, 0, 0}, pm2 = {0, 0, 0}, pm3 = {0, 0, 0};
            u32 pminp = 0xffffffffu;
            const u32 nchunks = ok ? (L + 63) / 64 : 0;
            for (u32 c = 0; c < nchunks; ++c) {
                const u32 t = c * 64 + (u32)lane;
                const u32 wd = w[c * 4 + ((u32)lane >> 4)];
                const u32 code = (wd >> (2 * ((u32)lane & 15))) & 3u;
                const u32x4 sd = *reinterpret_cast<const u32x4 *>(&s_seed[code * 4]);
                u32 Al = sd.x, Ah = sd.y, Bl = sd.z, Bh = sd.w;
                const lmask inread = __builtin_amdgcn_ballot_w64(t < L);
                Al = sel(inread, Al, 0u);  // bases past the end contribute nothing (their k-mers are never valid anyway)
                Ah = sel(inread, Ah, 0u);
                Bl = sel(inread, Bl, 0u);
                Bh = sel(inread, Bh, 0u);
                ror64v(Al, Ah, rA5, swA);
                ror64v(Bl, Bh, rB5, swB);
                const u32 QAl = wave_xor_scan(Al) ^ cAl, QAh = wave_xor_scan(Ah) ^ cAh, QBl = wave_xor_scan(Bl) ^ cBl, QBh = wave_xor_scan(Bh) ^ cBh;
                cAl = (u32)__builtin_amdgcn_readlane((int)QAl, 63);
                cAh = (u32)__builtin_amdgcn_readlane((int)QAh, 63);
                cBl = (u32)__builtin_amdgcn_readlane((int)QBl, 63);
                cBh = (u32)__builtin_amdgcn_readlane((int)QBh, 63);
                // k-mer ending at base t: index i = t - k + 1
                u32 fl = QAl ^ look_back_k(pQAl, QAl, pvK, adK), fh = QAh ^ look_back_k(pQAh, QAh, pvK, adK);
                u32 rl = QBl ^ look_back_k(pQBl, QBl, pvK, adK), rh = QBh ^ look_back_k(pQBh, QBh, pvK, adK);
                pQAl = QAl;
                pQAh = QAh;
                pQBl = QBl;
                pQBh = QBh;
                ror64v(fl, fh, rB5, swB);  // rol by l
                ror64v(rl, rh, rR5, swR);  // ror by i mod 64
                const u32 i = t - (u32)k + 1;  // wraps for t < k-1: then "i < nk" is false
                const lmask rev = lt64(rl, rh, fl, fh);
                HP h;
                h.lo = sel(rev, rl, fl);
                h.hi = sel(rev, rh, fh);
                h.p = (sel01(rev) << 15) | (i & 0x7fffu);
                // leftmost minimum of the W = 11 k-mers ending here
                const HP m1 = lmin(look_back3<1>(ph, h, pv1, ad1), h);       // 2: [i-1, i]
                const HP m2 = lmin(look_back3<2>(pm1, m1, pv2, ad2), m1);    // 4: [i-3, i]
                const HP m3 = lmin(look_back3<4>(pm2, m2, pv4, ad4), m2);    // 8: [i-7, i]
                const HP m = lmin(look_back3<3>(pm3, m3, pv3, ad3), m2);     // 11: [i-10, i-3] + [i-3, i]
                ph = h;
                pm1 = m1;
                pm2 = m2;
                pm3 = m3;
                const u32 before = look_back<1>(pminp, m.p, pv1, ad1);  // the previous window's minimum
                pminp = m.p;
                const lmask valid = __builtin_amdgcn_ballot_w64(i < nk && i >= (u32)(W - 1));
                const lmask e = valid & (__builtin_amdgcn_ballot_w64(m.p != before) | __builtin_amdgcn_ballot_w64(i == (u32)(W - 1)));
                const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(e >> 32), __builtin_amdgcn_mbcnt_lo((u32)e, 0));
                const u32 ne = (u32)__builtin_popcountll(e);
                if ((e >> lane) & 1) {
                    const u32 slot = cnt + rank;
                    if (slot < limit) {
                        a.hash[base + slot] = ((u64)m.hi << 32) | m.lo;
                        a.pos[base + slot] = (m.p & 0x7fffu) | ((m.p & 0x8000u) << 16);
                    }
                }
                cnt += ne;
                if (c == 0) {
                    // BSK_ST_FIRST_WINDOW_TIE (suffix_min_pass): the first W hashes are k-mers 0..W-1, in lanes k-1 .. k+W-2 of the first
                    // chunk (the planner keeps k + W <= 65 for this experiment); a short serial pass over broadcast values
                    u64 mn = 0;
                    bool dup = false;
                    for (int q = W - 1; q >= 0; --q) {
                        const int ln = k - 1 + q;
                        const u64 hv = ((u64)(u32)__builtin_amdgcn_readlane((int)h.hi, ln) << 32) | (u32)__builtin_amdgcn_readlane((int)h.lo, ln);
                        if (q == W - 1 || hv < mn) {
                            mn = hv;
                            dup = false;
                        } else if (hv == mn) {
                            dup = true;
                        }
                        tie |= dup ? 1u : 0u;
                    }
                }
            }
            if (cnt <= limit) break;
            u64 ob = 0;
            if (lane == 0) ob = atomicAdd(a.total + 1, (u64)cnt);
            ob = wave_bcast_u64(ob, 0);
            if (pass == 1 || ob + cnt > a.ovf_cap) {  // overflow region too small: flagged, the host re-runs with a larger one
                if (lane == 0) atomicOr(&a.ticket[1], 1u);
                cnt = 0;
                break;
            }
            base = a.ovf_base + ob;
            limit = cnt;
          }
            if ((u32)lane == j) {
                my_cnt = cnt;
                my_first = base;
                my_flags = (ok ? BSK_ST_OK : BSK_ST_SHORT) | ((tie & 1u) && ok ? BSK_ST_FIRST_WINDOW_TIE : 0);
            }
        }
        const u64 r = r0 + lane;
        if (r < a.n) {
            a.refs[r] = (my_first << 24) | my_cnt;
            u8 sbyte = (u8)my_flags;
            if ((my_flags & BSK_ST_CODE_MASK) == BSK_ST_OK && a.rflags) sbyte |= a.rflags[r];
            a.status[r] = sbyte;
        }
    }
}

}  // namespace bsk
