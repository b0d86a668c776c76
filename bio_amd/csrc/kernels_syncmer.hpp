// kernels_syncmer.hpp -- the reference's window-bounded closed syncmer (sketch.go:312-477),
// specialised on W = k - s (compile time), 2-bit input, one read per lane.
//
// Closed form: for idx in [0, end], end = L-2k+s+1, let mI be the leftmost
// argmin of the canonical s-mer hashes at s-mer positions [idx, idx+2W-1];
// b = mI if mI-idx < W else mI-W; every distinct b <= end is emitted once, in increasing
// order, as the canonical k-mer hash at b.
//
// ONE fused pass per read:
//   * the s-mer stream runs 2W-1 positions ahead of idx; its sliding minimum over W s-mers
//     (block decomposition in VGPRs, exactly as in kernels_fast.hpp) gives M[j] = leftmost
//     min of s-mers [j, j+W-1]; the 2W window of idx is min(M[idx], M[idx+W]) with the left
//     half winning ties, and "left half" / "right half" is exactly the prefix / suffix rule;
//     M[idx] was produced W steps earlier at the same block offset, so it sits in D[o];
//   * the selected position b lies in [idx, idx+W-1]: bit 31-(b-idx) of a per-lane pending
//     mask is set; the mask shifts left once per step, so its sign bit says "position idx is
//     selected" exactly when the k-mer stream (which runs AT idx) has that k-mer's hash --
//     this is the reference's "emit when idx reaches it" queue (sketch.go:424-475) without a
//     queue, de-duplication is the OR, and selections beyond `end` never reach bit 0;
//   * the (hash, idx|strand) of every step is stored to the lane's next LDS staging slot and
//     the slot advances only on a selection (branch-free); copy-out as in kernels_fast.hpp.
#pragma once
#include "kernels_fast.hpp"

namespace bsk {

template <int W, int CAP>
struct SynLds {
    static constexpr int ROW = 65;
    static constexpr int TABK = 0;    // k-mer update table (20 x 16 B)
    static constexpr int TABS = 512;  // s-mer update table
    static constexpr int SH = 1024;   // u64 [(CAP+1)*65]
    static constexpr int SP = SH + (CAP + 1) * ROW * 8;  // u16 [(CAP+1)*65]
    static constexpr int EXCL = SP + (((CAP + 1) * ROW * 2 + 15) & ~15);
    static constexpr int HEADS = EXCL + 256;
    static constexpr int NZ = HEADS + (CAP + 1) * 8;
    static constexpr int TOTAL = NZ + 64;
};

// SynLds + the four byte tables of the ASCII path (build_bytetabs for k and for s): k_syncmer_fast<W, false, true>, the side launch over
// the reads of a mixed batch that hold a non-ACGT letter (round 5; the general per-lane kernel before)
template <int W, int CAP>
struct SynLdsA : SynLds<W, CAP> {
    static constexpr int TINK = (SynLds<W, CAP>::TOTAL + 15) & ~15;
    static constexpr int TOUTK = TINK + 4096;
    static constexpr int TINS = TOUTK + 4096;
    static constexpr int TOUTS = TINS + 4096;
    static constexpr int TOTAL = TOUTS + 4096;
};

// 32 consecutive 2-bit codes starting at base position p0 (p0 >= 0): lo = codes 0..15, hi = codes 16..31
struct Codes32 {
    u32 lo, hi;
    __device__ __forceinline__ void load(const u32 *__restrict__ w, u32 p0) {
        const u32 wi = p0 >> 4, sh = (p0 & 15) * 2;
        const u32 w0 = w[wi], w1 = w[wi + 1], w2 = w[wi + 2];
        lo = __builtin_amdgcn_alignbit(w1, w0, sh);
        hi = __builtin_amdgcn_alignbit(w2, w1, sh);
    }
    // the same in two halves: the three words are requested one block ahead (issue) and cut to codes when the block starts
    // (finish) -- loaded where they are used, every block began with four dependent L2 round trips
    struct Raw {
        u32 a, b, c;
    };
    static __device__ __forceinline__ Raw issue(const u32 *__restrict__ w, u32 p0) {
        const u32 wi = p0 >> 4;
        Raw r;
        r.a = w[wi];
        r.b = w[wi + 1];
        r.c = w[wi + 2];
        return r;
    }
    __device__ __forceinline__ void finish(const Raw &r, u32 p0) {
        const u32 sh = (p0 & 15) * 2;
        lo = __builtin_amdgcn_alignbit(r.b, r.a, sh);
        hi = __builtin_amdgcn_alignbit(r.c, r.b, sh);
    }
    template <int O>
    __device__ __forceinline__ u32 in_off() const {  // code O as table byte offset (<< 4)
        constexpr int o = O & 15;
        const u32 v = O < 16 ? lo : hi;
        return (o >= 2 ? (v >> (2 * o - 4)) : (v << (4 - 2 * o))) & 0x30u;
    }
    template <int O>
    __device__ __forceinline__ u32 out_off() const {  // code O as table byte offset (<< 6)
        constexpr int o = O & 15;
        const u32 v = O < 16 ? lo : hi;
        return (o >= 3 ? (v >> (2 * o - 6)) : (v << (6 - 2 * o))) & 0xC0u;
    }
};

// ASC: the sequence is ASCII (`ab`: its first byte); a step's table rows are tin[incoming byte] ^ tout[outgoing byte] (SynLdsA)
template <int W, int CAP, bool DIRECT, bool ASC = false>
struct FastSyn {
    typedef typename std::conditional<ASC, SynLdsA<W, CAP>, SynLds<W, CAP>>::type LY;
    const u32 *__restrict__ w;
    const u8 *__restrict__ ab;
    LDSQ char *lds;
    int k, s, lane;
    u32 end_plus1;  // number of windows of this lane (end + 1), 0 if the read is short
    u64 *__restrict__ ghash;
    u32 *__restrict__ gpos;
    u64 gbase;
    // state
    u32 kfl, kfh, krl, krh, sfl, sfh, srl, srh;
    HV S[W], D[W], P;
    u32 pend, slot, tie, cnt;
    Codes32::Raw nsin, nsout, nkin, nkout;  // packed words of the NEXT steady block's four code streams
    u32 Al, Ah;  // first-window tie flag: the smallest suffix minimum of block 0 that occurs twice inside block 0 (~0: none)
    lmask tm;

    __device__ __forceinline__ void rollk(u32x4 x) {
        const u32 a = __builtin_amdgcn_alignbit(kfl, kfh, 31), b = __builtin_amdgcn_alignbit(kfh, kfl, 31);
        const u32 c = __builtin_amdgcn_alignbit(krh, krl, 1), d = __builtin_amdgcn_alignbit(krl, krh, 1);
        kfl = a ^ x.x;
        kfh = b ^ x.y;
        krl = c ^ x.z;
        krh = d ^ x.w;
    }
    __device__ __forceinline__ void rolls(u32x4 x) {
        const u32 a = __builtin_amdgcn_alignbit(sfl, sfh, 31), b = __builtin_amdgcn_alignbit(sfh, sfl, 31);
        const u32 c = __builtin_amdgcn_alignbit(srh, srl, 1), d = __builtin_amdgcn_alignbit(srl, srh, 1);
        sfl = a ^ x.x;
        sfh = b ^ x.y;
        srl = c ^ x.z;
        srh = d ^ x.w;
    }
    __device__ __forceinline__ u32x4 tabk(u32 off) const { return *reinterpret_cast<LDSQ const u32x4 *>(lds + LY::TABK + off); }
    __device__ __forceinline__ u32x4 tabs(u32 off) const { return *reinterpret_cast<LDSQ const u32x4 *>(lds + LY::TABS + off); }

    // MODE 0: s-mer block 0 (priming). MODE 1: block 1 (first 2W window completes at its last offset).
    // MODE 2: steady state (every offset is a fused step).
    template <int MODE, int O>
    __device__ __forceinline__ void step(u32 i0, const Codes32 &sin, const Codes32 &sout, const Codes32 &kin, const Codes32 &kout) {
        // ---- s-mer i_s = i0 + O ----
        if constexpr (ASC) {
            u32x4 x = *reinterpret_cast<LDSQ const u32x4 *>(lds + LY::TINS + ((u32)ab[i0 + (u32)O + (u32)s - 1u] << 4));
            if (!(MODE == 0 && O == 0)) {
                const u32x4 y = *reinterpret_cast<LDSQ const u32x4 *>(lds + LY::TOUTS + ((u32)ab[i0 + (u32)O - 1u] << 4));
                x = (u32x4){x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w};
            }
            rolls(x);
        } else {
        const u32 so = (MODE == 0 && O == 0) ? 0x100u : sout.template out_off<O>();
        rolls(tabs(sin.template in_off<O>() | so));
        }
        const lmask srev = lt64(srl, srh, sfl, sfh);
        HV v;
        v.lo = sel(srev, srl, sfl);
        v.hi = sel(srev, srh, sfh);
        v.p = i0 + O;  // s-mer position (wave-uniform value in a VGPR)
        if (O == 0) P = v;
        else P = selv(lt64(v.lo, v.hi, P.lo, P.hi), v, P);
        if (MODE >= 1 || O == W - 1) {
            HV M = P;  // leftmost min of s-mers [i_s-W+1, i_s]
            if constexpr (MODE >= 1 && O != W - 1) M = selv(lt64(P.lo, P.hi, S[O + 1].lo, S[O + 1].hi), P, S[O + 1]);
            if (MODE == 2 || (MODE == 1 && O == W - 1)) {
                // ---- fused step: idx = i_s - 2W + 1 ----
                const u32 idx = i0 + O - (2 * W - 1);
                const lmask right = lt64(M.lo, M.hi, D[O].lo, D[O].hi);  // strict: the left half wins ties
                const u32 b = sel(right, M.p - (u32)W, D[O].p);          // sketch.go:413-420
                pend |= 0x80000000u >> (b - idx);  // bit 31 = position idx: "selected" is the sign bit
                // k-mer at idx
                if constexpr (ASC) {
                    u32x4 x = *reinterpret_cast<LDSQ const u32x4 *>(lds + LY::TINK + ((u32)ab[idx + (u32)k - 1u] << 4));
                    if (MODE != 1) {
                        const u32x4 y = *reinterpret_cast<LDSQ const u32x4 *>(lds + LY::TOUTK + ((u32)ab[idx - 1u] << 4));
                        x = (u32x4){x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w};
                    }
                    rollk(x);
                } else {
                const u32 ko = (MODE == 1) ? 0x100u : kout.template out_off<O>();
                rollk(tabk(kin.template in_off<O>() | ko));
                }
                const lmask krev = lt64(krl, krh, kfl, kfh);
                const u32 hl = sel(krev, krl, kfl), hh = sel(krev, krh, kfh);
                const u32 ps = (sel01(krev) << 15) | idx;
                lmask e = __builtin_amdgcn_ballot_w64((int)pend < 0) & __builtin_amdgcn_ballot_w64(idx < end_plus1);
                pend <<= 1;
                if (!DIRECT) {
                    const u32 spare = (u32)(CAP * LY::ROW + lane) * 8u;
                    const u32 addr = slot < spare ? slot : spare;  // a full lane scribbles on the spare row
                    *reinterpret_cast<LDSQ u64 *>(lds + LY::SH + addr) = ((u64)hh << 32) | hl;
                    *reinterpret_cast<LDSQ u16 *>(lds + LY::SP + (addr >> 2)) = (u16)ps;
                    slot = sel(e, slot + (u32)(LY::ROW * 8), slot);
                } else if ((e >> lane) & 1) {
                    const u32 c = (slot - (u32)lane * 8u) / (u32)(LY::ROW * 8);
                    ghash[gbase + c] = ((u64)hh << 32) | hl;
                    gpos[gbase + c] = (ps & 0x7fffu) | ((ps & 0x8000u) << 16);
                    slot += (u32)(LY::ROW * 8);
                }
            }
            D[O] = M;
        }
        S[O] = v;
    }

    template <int MODE, int O>
    __device__ __forceinline__ void steps(u32 i0, const Codes32 &sin, const Codes32 &sout, const Codes32 &kin, const Codes32 &kout) {
        if constexpr (O < W) {
            step<MODE, O>(i0, sin, sout, kin, kout);
            steps<MODE, O + 1>(i0, sin, sout, kin, kout);
        }
    }

    __device__ __forceinline__ void prefetch(u32 i0) {  // words of the steady block starting at s-mer i0
        const u32 idx0 = i0 - (2 * W - 1);
        nsin = Codes32::issue(w, i0 + (u32)s - 1);
        nsout = Codes32::issue(w, i0 - 1);
        nkin = Codes32::issue(w, idx0 + (u32)k - 1);
        nkout = Codes32::issue(w, idx0 - 1);
    }

    template <int MODE>
    __device__ __forceinline__ void block(u32 i0) {
        Codes32 sin, sout, kin, kout;
        if constexpr (ASC) {
            sin.lo = sin.hi = sout.lo = sout.hi = kin.lo = kin.hi = kout.lo = kout.hi = 0;  // (unused: the steps read bytes)
        } else
        if (MODE == 2) {
            const u32 idx0 = i0 - (2 * W - 1);
            sin.finish(nsin, i0 + (u32)s - 1);
            sout.finish(nsout, i0 - 1);
            kin.finish(nkin, idx0 + (u32)k - 1);
            kout.finish(nkout, idx0 - 1);
            prefetch(i0 + W);  // (past the last block: inside the buffer's slack, never used)
        } else {
            sin.load(w, i0 + (u32)s - 1);
            sout.load(w, i0 ? i0 - 1 : 0);
        }
        if (MODE == 0) {  // block 0: offset o >= 1 sees base o-1 (offset 0 takes the "nothing leaves" row)
            const u64 v = (((u64)sout.hi << 32) | sout.lo) << 2;
            sout.lo = (u32)v;
            sout.hi = (u32)(v >> 32);
        }
        if (MODE == 2 || ASC) {
        } else if (MODE == 1) {
            prefetch(2 * W);
            // only offset W-1 is a fused step: idx = 0, incoming base k-1; place it at code index W-1
            kin.load(w, (u32)k - 1);
            // shift so that code 0 of the load appears at index W-1
            const u64 v = (((u64)kin.hi << 32) | kin.lo) << (2 * (W - 1));
            kin.lo = (u32)v;
            kin.hi = (u32)(v >> 32);
            kout.lo = kout.hi = 0;
        } else {
            kin.lo = kin.hi = kout.lo = kout.hi = 0;
        }
        steps<MODE, 0>(i0, sin, sout, kin, kout);
        // BSK_ST_FIRST_WINDOW_TIE over the first 2W s-mers (blocks 0 and 1; definition: kernels_fast.hpp, suffix_min_pass):
        // "the minimum of s-mers [q, 2W) occurs twice".  For q in block 1 that is the chain of block 1's own pass.  For q in
        // block 0, with S0[q] = min of block 0 from q and T = min of block 1:  S0[q] == T,  or  S0[q] < T and S0[q] occurs
        // twice inside block 0.  Block 0 therefore leaves A = the smallest such S0[q] (S0 is non-increasing towards q = 0, so
        // the last q of the descending pass wins) and parks its W suffix minima in the lane's staging rows 1..W, which hold no
        // tuple before the end of block 1; block 1 compares both with T.  3W instructions per read instead of W(2W-1) compares.
        if (MODE == 0 && !DIRECT) {
            lmask dup = 0;
#pragma unroll
            for (int q = W - 2; q >= 0; --q) {
                const lmask lt = lt64(S[q + 1].lo, S[q + 1].hi, S[q].lo, S[q].hi);
                dup = eq64(S[q + 1].lo, S[q + 1].hi, S[q].lo, S[q].hi) | (lt & dup);
                S[q] = selv(lt, S[q + 1], S[q]);
                if (dup) {  // wave-uniform and rare (a tie inside 2W s-mers: ~1 % of reads)
                    Al = sel(dup, S[q].lo, Al);
                    Ah = sel(dup, S[q].hi, Ah);
                }
            }
#pragma unroll
            for (int q = 0; q < W; ++q)
                *reinterpret_cast<LDSQ u64 *>(lds + LY::SH + (u32)((q + 1) * LY::ROW + lane) * 8u) = ((u64)S[q].hi << 32) | S[q].lo;
        } else if (MODE == 1 && !DIRECT) {
            suffix_min_pass<W, true>(S, tm);
            tm |= lt64(Al, Ah, S[0].lo, S[0].hi);
#pragma unroll
            for (int q = 0; q < W; ++q) {
                const u64 x = *reinterpret_cast<LDSQ const u64 *>(lds + LY::SH + (u32)((q + 1) * LY::ROW + lane) * 8u);
                tm |= eq64((u32)x, (u32)(x >> 32), S[0].lo, S[0].hi);
            }
        } else {
            lmask t0 = 0;
            suffix_min_pass<W, false>(S, t0);
        }
    }

    // ns_max: wave maximum of the number of s-mers (L - s + 1)
    __device__ __forceinline__ void run(u32 ns_max) {
        kfl = kfh = krl = krh = sfl = sfh = srl = srh = 0;
        pend = 0;
        tie = 0;
        tm = 0;
        Al = Ah = 0xffffffffu;
        slot = (u32)lane * 8u;
        if constexpr (ASC) {  // warm-ups from the bytes: tin rows alone (nothing leaves)
            for (int t = 0; t < s - 1; ++t) rolls(*reinterpret_cast<LDSQ const u32x4 *>(lds + LY::TINS + ((u32)ab[t] << 4)));
            for (int t = 0; t < k - 1; ++t) rollk(*reinterpret_cast<LDSQ const u32x4 *>(lds + LY::TINK + ((u32)ab[t] << 4)));
        } else {
        // warm-ups: four table rows in flight per trip (one row per trip exposes the LDS latency k + s - 2 times per read)
        for (int t0 = 0; t0 < s - 1; t0 += 16) {  // s-mer warm-up
            const u32 word = w[t0 >> 4];
            const int nb = (s - 1 - t0) < 16 ? (s - 1 - t0) : 16;
            int j = 0;
            for (; j + 4 <= nb; j += 4) {
                const u32 sub = word >> (2 * j);
                const u32x4 x0 = tabs(256 + ((sub & 3) << 4)), x1 = tabs(256 + ((sub & 0xc) << 2)), x2 = tabs(256 + (sub & 0x30)),
                            x3 = tabs(256 + ((sub & 0xc0) >> 2));
                rolls(x0);
                rolls(x1);
                rolls(x2);
                rolls(x3);
            }
            for (; j < nb; ++j) rolls(tabs(256 + (((word >> (2 * j)) & 3) << 4)));
        }
        for (int t0 = 0; t0 < k - 1; t0 += 16) {  // k-mer warm-up
            const u32 word = w[t0 >> 4];
            const int nb = (k - 1 - t0) < 16 ? (k - 1 - t0) : 16;
            int j = 0;
            for (; j + 4 <= nb; j += 4) {
                const u32 sub = word >> (2 * j);
                const u32x4 x0 = tabk(256 + ((sub & 3) << 4)), x1 = tabk(256 + ((sub & 0xc) << 2)), x2 = tabk(256 + (sub & 0x30)),
                            x3 = tabk(256 + ((sub & 0xc0) >> 2));
                rollk(x0);
                rollk(x1);
                rollk(x2);
                rollk(x3);
            }
            for (; j < nb; ++j) rollk(tabk(256 + (((word >> (2 * j)) & 3) << 4)));
        }
        }
        block<0>(0);
        block<1>(W);  // unconditional: a lane that is not short has at least 2W s-mers (sketch.go:149), and a conditional block leaves
                      // D[] undefined on one path -- the compiler then keeps last unit's registers alive across the unit loop
        for (u32 i0 = 2 * W; i0 < ns_max; i0 += W) block<2>(i0);
        tie = (u32)((tm >> lane) & 1);
        cnt = (slot - (u32)lane * 8u) / (u32)(LY::ROW * 8);
    }
};


// LIST: the READS k_syncmer_pk listed in a.fixlist (u32 read numbers, one segment per workgroup of the main launch: list_append) -- reads in
// which two equal 27-bit s-mer keys met in a min operation, or whose staging column filled up -- 64 per wavefront, on this kernel's
// 64-bit machine; their tuples go to the overflow region.
// ASC (round 5): the side launch of a mixed batch -- the reads a.subset names, from their ASCII bytes, unit slabs in [a.out_base, a.ovf_base),
// units with a read over its 28 tuples in [a.ovf_base, + a.ovf_cap) as in the main launch
template <int W, bool LIST = false, bool ASC = false>
__global__ __launch_bounds__(64, (ASC ? 1 : 2)) void k_syncmer_fast(KArgs a) {  // (ASC: 36 KB of LDS, four waves per CU)  // 2 waves per SIMD: at most 256 VGPRs.  One wave with 512 registers
                                                                    // removes the W >= 13 spills but runs 1.5x slower (432 -> 284 Gbases/s at W = 20)
    constexpr int CAP = BSK_SYN_CAP;
    typedef typename std::conditional<ASC, SynLdsA<W, CAP>, SynLds<W, CAP>>::type LY;
    __shared__ __attribute__((aligned(16))) char lds[LY::TOTAL];
    LDSQ char *const ldsq = (LDSQ char *)lds;
    const int lane = lane_id();
    if constexpr (ASC) {
        build_bytetabs(reinterpret_cast<uint4 *>(lds + SynLdsA<W, CAP>::TINK), reinterpret_cast<uint4 *>(lds + SynLdsA<W, CAP>::TOUTK), a.k, lane);
        build_bytetabs(reinterpret_cast<uint4 *>(lds + SynLdsA<W, CAP>::TINS), reinterpret_cast<uint4 *>(lds + SynLdsA<W, CAP>::TOUTS), a.s, lane);
    } else {
        build_xtab(reinterpret_cast<uint4 *>(lds + LY::TABK), a.k, lane);
        build_xtab(reinterpret_cast<uint4 *>(lds + LY::TABS), a.s, lane);
    }
    __syncthreads();
    const u64 slab = (u64)64 * CAP;
    u64 d_next = 0;
    bool pre = false;
    // LIST: this workgroup's segment of the list (list_append, kernels_generic.hpp)
    const u32 lseg = LIST ? a.fixcap / a.list_grid : 0u;
    const u32 *const flist = reinterpret_cast<const u32 *>(a.fixlist);
    for (u32 sg = LIST ? blockIdx.x : 0u; sg < (LIST ? a.list_grid : 1u); sg += LIST ? gridDim.x : 1u) {
    const u32 nlist = LIST ? (flist[sg] < lseg ? flist[sg] : lseg) : 0u;
    const u32 *const rlist = LIST ? flist + a.list_grid + sg * lseg : nullptr;
    const u32 TK = ASC ? 1u : (a.tk ? a.tk : 8u);  // units per ticket (the side launch's few units are latency: one per wavefront; KArgs::tk: fewer for small batches)
    for (u32 unit = LIST ? 0u : next_ticket(a.ticket, lane) * TK, uend = unit + TK; LIST ? unit < (nlist + 63u) / 64u : unit < a.nunits; ++unit, ({
             if (LIST) {
             } else if (unit == uend) {
                 unit = next_ticket(a.ticket, lane) * TK;
                 uend = unit + TK;
             }
         })) {
        u64 r = (u64)unit * 64 + lane;
        if (LIST) r = r < nlist ? (u64)rlist[r] : ~0ULL;
        if (ASC) r = r < a.nsub ? (u64)a.subset[r] : ~0ULL;
        // the next unit's descriptors are loaded one unit ahead (a load issued here waits for the copy-out stores to drain)
        u64 d = 0, off = 0, L = 0, ro = r;
        if constexpr (ASC) {
            if (r < a.n) ascii_span(a, r, off, L);  // (byte offset and length of the read's ASCII)
        } else {
            d = (pre && !LIST) ? d_next : (r < a.n ? a.desc[r] : 0);
            off = d >> 24;
            L = desc_len(a, d);
            ro = out_index(a, r, d);  // (length-binned batches: the read's own place in its chunk)
            pre = !LIST && unit + 1 != uend && unit + 1 < a.nunits;
            if (pre) d_next = r + 64 < a.n ? a.desc[r + 64] : 0;
        }
        const long long Lorig = (long long)L - a.circ_ext;
        const bool ok = r < a.n && Lorig >= 0 && Lorig >= 2LL * a.k - a.s - 1 && L >= (u64)a.k;  // sketch.go:149
        const u32 nwin = ok ? (u32)(L - 2 * (u64)a.k + a.s + 2) : 0u;                             // end + 1
        const u32 ns = ok ? (u32)(L - a.s + 1) : 0u;
        const u32 ns_max = wave_max_u32(ns);
        u32 cnt = 0, tie = 0;
        if (ns_max) {
            FastSyn<W, CAP, false, ASC> fs;
            fs.w = a.words + off;
            fs.ab = ASC ? a.ascii + off : nullptr;
            fs.lds = ldsq;
            fs.k = a.k;
            fs.s = a.s;
            fs.lane = lane;
            fs.end_plus1 = nwin;
            fs.run(ns_max);
            cnt = fs.cnt;
            tie = ok ? fs.tie : 0;
        }
        const u32 incl = wave_incl_scan_u32(cnt, lane);
        const u32 excl = incl - cnt;
        const u32 T = wave_bcast_u32(incl, 63);
        const bool any_over = __builtin_amdgcn_ballot_w64(cnt > (u32)CAP) != 0;
        u64 base = (ASC ? a.out_base : 0) + (u64)unit * slab;
        u64 ob = 0;
        if (LIST || any_over) {
            if (lane == 0) ob = atomicAdd(a.total + 1, (u64)T);
            ob = wave_bcast_u64(ob, 0);
        }
        if (LIST && ob + T <= a.ovf_cap) base = a.ovf_base + ob;
        if (!any_over && (!LIST || ob + T <= a.ovf_cap)) {
            fast_copyout<LY, true, CAP, 4>(lds, lane, cnt, excl, T, base, a);
        } else {
            if (ob + T <= a.ovf_cap) {
                base = a.ovf_base + ob;
                FastSyn<W, CAP, true, ASC> fs;
                fs.w = a.words + off;
                fs.ab = ASC ? a.ascii + off : nullptr;
                fs.lds = ldsq;
                fs.k = a.k;
                fs.s = a.s;
                fs.lane = lane;
                fs.end_plus1 = nwin;
                fs.ghash = a.hash;
                fs.gpos = a.pos;
                fs.gbase = base + excl;
                fs.run(ns_max);
            } else {
                cnt = 0;
                if (lane == 0) atomicOr(&a.ticket[1], 1u);
            }
        }
        if (r < a.n) {
            a.refs[ro] = ((base + excl) << 24) | cnt;
            u8 sbyte = ok ? BSK_ST_OK : BSK_ST_SHORT;
            if (tie && ok) sbyte |= BSK_ST_FIRST_WINDOW_TIE;
            if (ok && a.rflags) sbyte |= a.rflags[r];
            a.status[ro] = sbyte;
        }
    }
    }
}

#ifdef BSK_IMPL_SYNCMER  // dispatch functions: compiled in the family's own translation unit
#ifndef BSK_SYN_WS  // (dev builds narrow the list: -D'BSK_SYN_WS(X)=X(20)')
#define BSK_SYN_WS(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24)
#endif
bool fast_syncmer_supported(int k, int s) {
    switch (k - s) {
#define X(WW) case WW:
        BSK_SYN_WS(X)
#undef X
        return true;
        default: return fast_syncmer_wide_supported(k - s);
    }
}
int fast_syncmer_blocks_per_cu(int w) {
    int nb = 0;
    hipError_t e = hipErrorInvalidValue;
    switch (w) {
#define X(WW) \
    case WW: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_syncmer_fast<WW>, 64, 0); break;
        BSK_SYN_WS(X)
#undef X
        default: return fast_syncmer_wide_blocks_per_cu(w);
    }
    if (e != hipSuccess || nb < 1) {
        (void)hipGetLastError();
        nb = 1;
    }
    return nb;
}
void fast_syncmer_launch(int w, int grid, hipStream_t stream, const KArgs &a) {
    switch (w) {
#define X(WW) \
    case WW: hipLaunchKernelGGL((k_syncmer_fast<WW>), dim3(grid), dim3(64), 0, stream, a); break;
        BSK_SYN_WS(X)
#undef X
        default: fast_syncmer_wide_launch(w, grid, stream, a); break;
    }
}

#endif  // BSK_IMPL_SYNCMER
#ifdef BSK_IMPL_SYNCMER_WIDE  // k_syncmer_wide.hip: k - s = 25..32 (round 5: the general kernel before, 25-106 Gbases/s at k=63 s=31), a
                              // translation unit of their own; the block of W steps reads 32 codes, the pending mask has 32 bits: W <= 32
#define BSK_SYN_WIDE_WS(X) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32)
bool fast_syncmer_wide_supported(int w) { return w >= 25 && w <= 32; }
int fast_syncmer_wide_blocks_per_cu(int w) {
    int nb = 0;
    hipError_t e = hipErrorInvalidValue;
    switch (w) {
#define X(WW) \
    case WW: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_syncmer_fast<WW>, 64, 0); break;
        BSK_SYN_WIDE_WS(X)
#undef X
        default: break;
    }
    if (e != hipSuccess || nb < 1) {
        (void)hipGetLastError();
        nb = 1;
    }
    return nb;
}
void fast_syncmer_wide_launch(int w, int grid, hipStream_t stream, const KArgs &a) {
    switch (w) {
#define X(WW) \
    case WW: hipLaunchKernelGGL((k_syncmer_fast<WW>), dim3(grid), dim3(64), 0, stream, a); break;
        BSK_SYN_WIDE_WS(X)
#undef X
        default: break;
    }
}
#endif  // BSK_IMPL_SYNCMER_WIDE
#ifdef BSK_IMPL_SYNCMER_ASCII  // k_syncmer_ascii.hip: the ASCII side launch's instantiations, a translation unit of their own
#ifndef BSK_SYN_WS
#define BSK_SYN_WS(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24)
#endif
void fast_syncmer_ascii_launch(int w, int grid, hipStream_t stream, const KArgs &a) {  // the side launch over a.subset, from ASCII
    switch (w) {
#define X(WW) \
    case WW: hipLaunchKernelGGL((k_syncmer_fast<WW, false, true>), dim3(grid), dim3(64), 0, stream, a); break;
        BSK_SYN_WS(X)
#undef X
        default: break;
    }
}
#endif  // BSK_IMPL_SYNCMER_ASCII


}  // namespace bsk
