// kernels_fast.hpp -- minimizer kernel specialised on the window size W (compile time).
//
// Mapping (same as kernels_generic.hpp): one read per lane, 64 reads ("unit") per
// wavefront, wave-uniform loop counters, true rolling ntHash per lane.  What is
// specialised here:
//   * the k-mer loop is unrolled by W: the sliding-window state (suffix minima of the
//     previous block of W k-mers, running prefix minimum of the current block) lives
//     in VGPRs with static indices -- no LDS/HBM traffic for the window;
//   * the 2-bit codes of the W incoming / W outgoing bases of a block are cut out of
//     the packed stream once per block (one v_alignbit each); the packed words of the
//     NEXT block are requested while the current block is hashed;
//   * per k-mer the only memory read is ONE 16-byte LDS fetch of the (outgoing,
//     incoming) -> (forward, reverse) update table; all W of them are issued at the top
//     of the block, ahead of the dependent hash chain;
//   * no inter-wavefront synchronisation at all: unit u owns the fixed slab
//     [u*64*CAP, (u+1)*64*CAP) of the tuple arrays, so its tuples go out with nothing to
//     wait for (a look-back chain over the ~1500 resident units cost 37 % of the run
//     time in the first version of this kernel);
//   * instruction selection follows the measured gfx950 integer issue rates
//     (scripts/ubench): and/or/xor/lshr/add are full rate, everything VOP3 is half
//     rate, and a VOP2 v_cndmask reading VCC costs ~5 half-rate slots -- so every
//     select is forced to the VOP3 form with an SGPR-pair mask.
// HBM traffic is the packed bases in and the selected (hash, pos|strand) tuples
// plus one 8-byte reference per read out; tuples leave LDS as whole 512-/256-byte rows.
#pragma once
#include <type_traits>
#include "fast_dispatch.hpp"

namespace bsk {

typedef u64 lmask;  // one bit per lane, lives in an SGPR pair
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define GLBQ __attribute__((address_space(1)))
typedef u32x4 u32x4_u __attribute__((aligned(1)));  // byte-aligned 16-byte global load (unaligned access mode of the HSA ABI)

// 16-byte non-temporal store (8-byte aligned): the outputs are written once and never read by the kernels; keeping them out
// of the L2's retained set leaves it to the inputs (k-mer codes 629 -> 663 Gbases/s, protein minimizer input re-fetches 5x -> 1.9x)
__device__ __forceinline__ void nt_store_u64x2(u64 *p, u64 a, u64 b) {
    typedef u64 u64x2v __attribute__((ext_vector_type(2), aligned(8)));
    __builtin_nontemporal_store((u64x2v){a, b}, reinterpret_cast<u64x2v *>(p));
}

// VOP3 compare -> SGPR pair, VOP3 select <- SGPR pair.  (v_cmp -> v_cndmask through an
// SGPR needs no software wait states on gfx9; the asm only pins the encoding.)
__device__ __forceinline__ lmask lt64(u32 alo, u32 ahi, u32 blo, u32 bhi) {
    return __builtin_amdgcn_ballot_w64((((u64)ahi << 32) | alo) < (((u64)bhi << 32) | blo));
}
__device__ __forceinline__ u32 sel(lmask m, u32 t, u32 f) {  // m ? t : f
    u32 r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(m));
    return r;
}

__device__ __forceinline__ u32 sel01(lmask m) {  // m ? 1 : 0 (both operands inline constants: no VGPR copies of wave-uniform values)
    u32 r;
    asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(r) : "s"(m));
    return r;
}

// tm |= lanes where a == b.  Compare and OR in one asm block: left to itself the scheduler issues all W(W-1)/2 compares of the
// first-window tie test first and spills their SGPR pairs to VGPR lanes (v_writelane + s_nop + v_readlane for every pair).
__device__ __forceinline__ void or_eq64(lmask &tm, u64 a, u64 b) {
    lmask t;
    asm("v_cmp_eq_u64_e64 %1, %2, %3\n\ts_or_b64 %0, %0, %1" : "+s"(tm), "=&s"(t) : "v"(a), "v"(b) : "scc");  // s_or writes SCC
}

__device__ __forceinline__ lmask eq64(u32 alo, u32 ahi, u32 blo, u32 bhi) {
    return __builtin_amdgcn_ballot_w64((((u64)ahi << 32) | alo) == (((u64)bhi << 32) | blo));
}

struct HV {  // hash (lo, hi) + pos|strand
    u32 lo, hi, p;
};
__device__ __forceinline__ HV selv(lmask m, HV t, HV f) {
    HV r;
    r.lo = sel(m, t.lo, f.lo);
    r.hi = sel(m, t.hi, f.hi);
    r.p = sel(m, t.p, f.p);
    return r;
}

// Suffix-minimum pass over the W values of a finished block: S[q] = min(S[q..W-1]), the older (left) element winning ties.
// FIRST (the block is the sequence's first sorted window, sketch.go:236): also evaluates BSK_ST_FIRST_WINDOW_TIE --
//   tie <=> two equal hashes h[t1] == h[t2], t1 < t2, inside the first window with nothing smaller in (t1, end of the window),
// i.e. "the minimum of [q, W) occurs twice" for some q: the only ties whose order after the reference's unstable first sort can
// ever reach buf[0] (two tied entries are both in the buffer while their value is its minimum only if nothing after t1 in the
// first window is smaller).  One equality test per element on top of the pass's own compare; the chain is scalar mask logic.
template <int W, bool FIRST>
__device__ __forceinline__ void suffix_min_pass(HV (&S)[W], lmask &tm) {
    lmask dup = 0;
#pragma unroll
    for (int q = W - 2; q >= 0; --q) {
        const lmask lt = lt64(S[q + 1].lo, S[q + 1].hi, S[q].lo, S[q].hi);
        if (FIRST) {
            // dup = eq | (lt & dup)  (equal: twice; S[q] smaller: unique so far);  tm |= dup.  One asm block per element, like
            // or_eq64: otherwise all W - 1 equality masks are computed first and live in spilled SGPR pairs (110 spills at w = 32)
            lmask t;
            asm("v_cmp_eq_u64_e64 %2, %3, %4\n\ts_and_b64 %0, %5, %0\n\ts_or_b64 %0, %2, %0\n\ts_or_b64 %1, %1, %0"
                : "+s"(dup), "+s"(tm), "=&s"(t)
                : "v"(((u64)S[q + 1].hi << 32) | S[q + 1].lo), "v"(((u64)S[q].hi << 32) | S[q].lo), "s"(lt)
                : "scc");  // s_and / s_or write SCC
        }
        S[q] = selv(lt, S[q + 1], S[q]);
    }
}

// LDS plan of one wavefront (bytes).  7 wavefronts per CU fit in the 160 KB.
#define LDSQ __attribute__((address_space(3)))
template <int CAP, bool POS16>
struct FLds {
    static constexpr int ROW = 65;  // slot(e, lane) = e*65 + lane: bank-conflict free for the per-lane
                                    // stores (equal e -> consecutive lanes) and for the copy-out reads
                                    // (one lane's consecutive e rotate through the banks).
                                    // Row CAP is a spare row: every step stores to the lane's NEXT slot (a non-selected
                                    // candidate is simply overwritten later), so a full lane scribbles on row CAP.
    static constexpr int PB = POS16 ? 2 : 4;
    static constexpr int TAB = 0;                                           // 20 x uint4 update table
    static constexpr int SH = 512;                                          // u64 [(CAP+1)*65]
    static constexpr int SP = SH + (CAP + 1) * ROW * 8;                     // u16|u32 [(CAP+1)*65]
    static constexpr int EXCL = SP + (((CAP + 1) * ROW * PB + 15) & ~15);   // u32 [64]
    static constexpr int HEADS = EXCL + 256;  // u64 [CAP+1]: bit j of word c: a lane's run starts at output 64c+j
    static constexpr int NZ = HEADS + (CAP + 1) * 8;  // u8 [64]: rank among non-empty lanes -> lane
    static constexpr int DST = NZ + 64;               // 512 bytes: the owner table of flush_groups / flush_last (kernels with per-read slabs)
    static constexpr int ROWS = CAP + 1;
    static constexpr int TAB2 = DST + 512;            // 16 x uint4: warm-up table of two bases (build_xtab2)
    static constexpr int TOTAL = TAB2 + 256;
};

// Paired staging columns (k_minimizer_fast): lanes l and l+32 share column l & 31 of R rows; the low lane fills it from row
// 0 upwards, the high lane from row R-1 downwards.  A column overflows when the two counts reach R together, so the
// capacity follows the SUM of two reads' counts (k=21 w=11, 150 bp: mean 44.3, sd 3.5) instead of the maximum of each
// (22.1, sd 2.45): 56 shared rows overflow as rarely as 2 x 32 private ones (1 % of units) and the wave's LDS drops
// from 22.3 KB to 19.2 KB -- 8 waves per CU instead of 7, and throughput is linear in the waves.
// Row R is the spare row for lanes that ran out of the column; the copy-out keeps its exclusive offsets there.
#define BSK_PAIR_ROWS 56
template <int R, bool POS16>
struct PLds {
    static constexpr int PR = R;    // rows of a column
    static constexpr int ROW = 33;  // 32 columns + 1: consecutive rows of a column rotate through the banks
    static constexpr int PB = POS16 ? 2 : 4;
    static constexpr int TAB = 0;
    static constexpr int SH = 512;                                         // u64 [(R+1)*33]
    static constexpr int SP = SH + (R + 1) * ROW * 8;                      // u16|u32 [(R+1)*33]
    static constexpr int HEADS = SP + (((R + 1) * ROW * PB + 15) & ~15);   // u64 [NHEADS]
    static constexpr int NHEADS = (32 * (R - 1)) / 64 + 2;                 // a unit holds at most 32*(R-1) tuples; the last word is never set
    static constexpr int NZ = HEADS + NHEADS * 8;                          // u8 [64]
    static constexpr int EXCL = SH + R * ROW * 8;                          // u32 [64] in the spare row (free once the pass is over)
    static constexpr int TAB2 = NZ + 64;                                   // 16 x uint4: warm-up table of two bases (build_xtab2)
    static constexpr int CTAB = TAB2 + 256;                                // u64 [64]: fast_copyout's owner table
    static constexpr int TOTAL = CTAB + 512;                               // 20 400 B: eight waves per CU use 163 200 of the 163 840 B
};
// FLds + the two byte tables of the ASCII path (build_bytetabs: tin[b], tout[b], 256 x 16 bytes each): k_minimizer_dense<W, false, true>,
// the side launch over the reads of a mixed batch that hold a non-ACGT letter
template <int CAP>
struct FLdsA : FLds<CAP, true> {
    static constexpr int TIN = (FLds<CAP, true>::TOTAL + 15) & ~15;
    static constexpr int TOUT = TIN + 4096;
    static constexpr int TOTAL = TOUT + 4096;
};
template <bool PAIR, int CAP, bool POS16, int PR>
struct MinLds {
    typedef FLds<CAP, POS16> type;
};
template <int CAP, bool POS16, int PR>
struct MinLds<true, CAP, POS16, PR> {
    typedef PLds<PR, POS16> type;
};

// RING: the lane's CAP+1 rows are a ring (k_minimizer_dense: no left-over moves after a flush); `send` = one row past the last.
// PR: rows of a paired column.  XCH: table rows fetched per chunk (0: all W up front for W <= 16, 4 beyond).
// ASC: the sequence is ASCII (`ab`: its first byte); a step's table row is tin[incoming byte] ^ tout[outgoing byte] (FLdsA)
template <int W, int CAP, bool POS16, bool DIRECT, bool PAIR = false, bool RING = false, int PR = BSK_PAIR_ROWS, int XCH = 0, bool ASC = false>
struct FastMin {
    typedef typename std::conditional<ASC, FLdsA<CAP>, typename MinLds<PAIR, CAP, POS16, PR>::type>::type LY;
    const u8 *__restrict__ ab;
    static constexpr u32 SBIT = POS16 ? 0x8000u : 0x80000000u;  // strand bit inside the staged pos word
    const u32 *__restrict__ w;
    LDSQ char *lds;
    static constexpr bool USE_T2 = PAIR && !RING && !DIRECT && XCH == 0;  // k_minimizer_fast's staged pass builds the two-base warm-up table
    int k, lane;
    u32 nk;
    u64 *__restrict__ ghash;
    u32 *__restrict__ gpos;
    u64 gbase;
    // rolling state
    u32 fl, fh_, rl, rh_;  // forward / reverse hash halves
    HV S[W];               // suffix minima of the previous block (then raw values of the current one)
    HV P;                  // running prefix minimum of the current block
    u32 prev, cnt, tie;
    u32 slot;                          // byte offset (from SH) of this lane's next staging slot = (cnt*65 + lane)*8
    u32 sstep, slim, sspare;           // PAIR: signed row stride, last byte offset inside the column, the lane's spare slot
    u32 send;                          // RING: slot offset one row past the lane's last row
    u32 in_lo, in_hi, out_lo, out_hi;  // packed words of the current block
    u32 in_h2, out_h2;                 // W > 16: a block spans up to three words
    u32 nku;                           // run(): the wave's largest window count (wave-uniform)
    u32x4 pw;                          // run(): the read's first four words, loaded by the caller (one unit ahead)

    __device__ __forceinline__ u32 first_word(u32 i) const {  // i is wave-uniform
        if (i < 4) return i == 0 ? pw.x : i == 1 ? pw.y : i == 2 ? pw.z : pw.w;
        return w[i];
    }
    __device__ __forceinline__ void load_block_words(u32 i0) {
        const u32 t0 = i0 + (u32)k - 1;
        in_lo = w[t0 >> 4];
        in_hi = w[(t0 >> 4) + 1];
        const u32 p0 = i0 ? i0 - 1 : 0;
        out_lo = w[p0 >> 4];
        out_hi = w[(p0 >> 4) + 1];
        if (W > 16) {
            in_h2 = w[(t0 >> 4) + 2];
            out_h2 = w[(p0 >> 4) + 2];
        }
    }
    __device__ __forceinline__ void roll(u32x4 x) {
        const u32 a = __builtin_amdgcn_alignbit(fl, fh_, 31), b = __builtin_amdgcn_alignbit(fh_, fl, 31);
        const u32 c = __builtin_amdgcn_alignbit(rh_, rl, 1), d = __builtin_amdgcn_alignbit(rl, rh_, 1);
        fl = a ^ x.x;
        fh_ = b ^ x.y;
        rl = c ^ x.z;
        rh_ = d ^ x.w;
    }

    __device__ __forceinline__ void roll2(u32x4 x) {  // two warm-up bases at once: fh' = rol(fh, 2) ^ F2, rh' = ror(rh, 2) ^ R2
        const u32 a = __builtin_amdgcn_alignbit(fl, fh_, 30), b = __builtin_amdgcn_alignbit(fh_, fl, 30);
        const u32 c = __builtin_amdgcn_alignbit(rh_, rl, 2), d = __builtin_amdgcn_alignbit(rl, rh_, 2);
        fl = a ^ x.x;
        fh_ = b ^ x.y;
        rl = c ^ x.z;
        rh_ = d ^ x.w;
    }

    // GUARD: some lane may reach CAP staged tuples inside this block (store must be bounded)
    template <bool FIRST, bool GUARD>
    __device__ __forceinline__ void block(u32 i0) {
        const u32 t0 = i0 + (u32)k - 1;
        const u32 cinb = __builtin_amdgcn_alignbit(in_hi, in_lo, (t0 & 15) * 2);  // code of slot o at bits [2o, 2o+2)
        u32 coutb;
        if (FIRST) coutb = out_lo << 2;  // slot 0: nothing leaves; slot o >= 1 sees base o-1
        else coutb = __builtin_amdgcn_alignbit(out_hi, out_lo, ((i0 - 1) & 15) * 2);
        u32 cinb2 = 0, coutb2 = 0;  // W > 16: codes of slots 16..31
        if (W > 16) {
            cinb2 = __builtin_amdgcn_alignbit(in_h2, in_hi, (t0 & 15) * 2);
            if (FIRST) coutb2 = __builtin_amdgcn_alignbit(out_hi, out_lo, 30);  // slot 16 sees base 15, ...
            else coutb2 = __builtin_amdgcn_alignbit(out_h2, out_hi, ((i0 - 1) & 15) * 2);
        }
        // Table rows: all W of the block are fetched up front for W <= 16; wider windows fetch them in chunks of XC, one chunk
        // ahead (4W VGPRs of rows in flight would push W >= 17 past 256 VGPRs, i.e. to one wave per SIMD or into spills).
        // Chunks of 4 measured best (w = 28: 734 Gbases/s against 679 / 540 / 390 with chunks of 6 / 8 / 12).
        constexpr int XC = XCH ? (XCH < W ? XCH : W) : (W > 16 ? 4 : W);
        u32x4 xs[W];
        auto fetch = [&](int o0) {
#pragma unroll
            for (int o = o0; o < o0 + XC && o < W; ++o) {  // byte offset into the table = out*64 + in*16 ; row "nothing leaves" = 256 + in*16
                const int q = o & 15;
                const u32 ci = o < 16 ? cinb : cinb2, co = o < 16 ? coutb : coutb2;
                const u32 a = (q >= 2 ? (ci >> (2 * q - 4)) : (ci << (4 - 2 * q))) & 0x30u;
                const u32 b = (FIRST && o == 0) ? 0x100u : ((q >= 3 ? (co >> (2 * q - 6)) : (co << (6 - 2 * q))) & 0xC0u);
                xs[o] = *reinterpret_cast<LDSQ const u32x4 *>(lds + LY::TAB + (a | b));
            }
        };
        if constexpr (ASC) {  // (bytes straight from memory, all of the block's at once; two table reads per step)
            u32 ib[W], ob[W];
#pragma unroll
            for (int o = 0; o < W; ++o) {
                ib[o] = ab[t0 + (u32)o];
                ob[o] = (FIRST && o == 0) ? 0u : ab[i0 + (u32)o - 1u];
            }
#pragma unroll
            for (int o = 0; o < W; ++o) {
                const u32x4 xi = *reinterpret_cast<LDSQ const u32x4 *>(lds + LY::TIN + (ib[o] << 4));
                if (FIRST && o == 0) {
                    xs[o] = xi;
                } else {
                    const u32x4 xo = *reinterpret_cast<LDSQ const u32x4 *>(lds + LY::TOUT + (ob[o] << 4));
                    xs[o] = (u32x4){xi.x ^ xo.x, xi.y ^ xo.y, xi.z ^ xo.z, xi.w ^ xo.w};
                }
            }
        } else {
        fetch(0);
        if (XC < W) fetch(XC);
        load_block_words(i0 + W);  // next block's words: in flight while this block is hashed
        }
        u32 vi = i0;               // k-mer index as a VGPR (selects need VGPR operands)
        const u32 spare = PAIR ? sspare : (u32)(CAP * LY::ROW + lane) * 8u;
#pragma unroll
        for (int o = 0; o < W; ++o) {
            if (XC < W && !FIRST && o && o % XC == 0) {
                // wide windows: the wave's last block ends at the chunk boundary past its last window (a 150-base read has 130
                // windows: five blocks of 32 would run 160 steps).  No block follows, so the suffix pass goes too.
                if (i0 + (u32)o >= nku) return;
            }
            if (!ASC && XC < W && o && o % XC == 0 && o + XC < W) {
                __builtin_amdgcn_sched_barrier(0);  // keep the next chunk's reads here (hoisted, they are all live at once again)
                fetch(o + XC);
            }
            roll(xs[o]);
            const lmask rev = lt64(rl, rh_, fl, fh_);
            HV v;
            v.lo = sel(rev, rl, fl);
            v.hi = sel(rev, rh_, fh_);
            v.p = (sel01(rev) << (POS16 ? 15 : 31)) | vi;  // vi is wave-uniform: one select + one v_lshl_or instead of two copies + a select
            if (o == 0) {
                P = v;
            } else {
                P = selv(lt64(v.lo, v.hi, P.lo, P.hi), v, P);
            }
            if (!FIRST || o == W - 1) {
                HV m = P;
                if (o != W - 1) {
                    const int o1 = o + 1 < W ? o + 1 : o;  // (folds when the loop is unrolled; keeps the dead last iteration inside the array)
                    m = selv(lt64(P.lo, P.hi, S[o1].lo, S[o1].hi), P, S[o1]);
                }
                lmask e = __builtin_amdgcn_ballot_w64(m.p != prev);
                e &= __builtin_amdgcn_ballot_w64(vi < nk);  // per-lane bound check (a wave-uniform variant without it did not pay for its code)
                prev = m.p;
                if (!DIRECT) {
                    // branch-free: every candidate is stored to the lane's next slot; the slot only advances when the
                    // candidate is a new selection, so anything else is overwritten (or left beyond the count)
                    u32 addr = slot;
                    if (GUARD) addr = slot < spare ? slot : spare;  // one v_min_u32 (a column left downwards wraps far above the spare row, one left upwards lands on it)
                    *reinterpret_cast<LDSQ u64 *>(lds + LY::SH + addr) = ((u64)m.hi << 32) | m.lo;
                    if (POS16) *reinterpret_cast<LDSQ u16 *>(lds + LY::SP + (addr >> 2)) = (u16)m.p;
                    else *reinterpret_cast<LDSQ u32 *>(lds + LY::SP + (addr >> 1)) = m.p;
                    u32 nxt = slot + (PAIR ? sstep : (u32)(LY::ROW * 8));
                    if (RING) nxt = sel(__builtin_amdgcn_ballot_w64(nxt == send), (u32)lane * 8u, nxt);
                    slot = sel(e, nxt, slot);
                } else {
                    if ((e >> lane) & 1) {
                        const u32 c = (slot - (u32)lane * 8u) / (u32)(LY::ROW * 8);
                        ghash[gbase + c] = ((u64)m.hi << 32) | m.lo;
                        gpos[gbase + c] = POS16 ? ((m.p & 0x7fffu) | ((m.p & 0x8000u) << 16)) : m.p;
                        slot += (u32)(LY::ROW * 8);
                    }
                }
            }
            S[o] = v;
            vi += 1;
        }
        if (FIRST) {  // (DIRECT too: k_minimizer_pk's exact re-run of a unit takes the flag from here)
            lmask tm = 0;
            suffix_min_pass<W, true>(S, tm);
            tie = (u32)((tm >> lane) & 1);
        } else {
            lmask tm = 0;
            suffix_min_pass<W, false>(S, tm);
        }
    }

    // state reset, warm-up over the first k-1 bases, and the words of block 0
    __device__ __forceinline__ void begin() {
        fl = fh_ = rl = rh_ = 0;
        prev = 0xffffffffu;
        tie = 0;
        nku = 0xffffffffu;  // callers that drive block() themselves never leave a block early
        slot = (u32)lane * 8u;
        const u32 col8 = (u32)(lane & 31) * 8u;
        const bool up = lane < 32;
        if (PAIR) {
            slim = (u32)((PR - 1) * LY::ROW * 8) + col8;
            sspare = (u32)(PR * LY::ROW * 8) + col8;
            slot = up ? col8 : slim;
            sstep = up ? (u32)(LY::ROW * 8) : (u32)(-(int)(LY::ROW * 8));
        }
        // warm-up: bases 0..k-2 enter, nothing leaves (table rows 16..19)
        for (int t0 = 0; t0 < k - 1; t0 += 16) {
            const u32 word = first_word((u32)t0 >> 4);
            const int nb = (k - 1 - t0) < 16 ? (k - 1 - t0) : 16;
            int j = 0;
            if (USE_T2) {
                for (; j + 8 <= nb; j += 8) {  // eight bases = four rows of the two-base table in flight
                    const u32 sub = word >> (2 * j);
                    const u32x4 x0 = *reinterpret_cast<LDSQ const u32x4 *>(lds + LY::TAB2 + ((sub & 0xf) << 4));
                    const u32x4 x1 = *reinterpret_cast<LDSQ const u32x4 *>(lds + LY::TAB2 + (sub & 0xf0));
                    const u32x4 x2 = *reinterpret_cast<LDSQ const u32x4 *>(lds + LY::TAB2 + ((sub & 0xf00) >> 4));
                    const u32x4 x3 = *reinterpret_cast<LDSQ const u32x4 *>(lds + LY::TAB2 + ((sub & 0xf000) >> 8));
                    roll2(x0);
                    roll2(x1);
                    roll2(x2);
                    roll2(x3);
                }
                for (; j + 2 <= nb; j += 2) roll2(*reinterpret_cast<LDSQ const u32x4 *>(lds + LY::TAB2 + (((word >> (2 * j)) & 0xf) << 4)));
            }
            for (; j + 4 <= nb; j += 4) {  // four table rows in flight (one row per trip exposes the LDS latency 20 times per read)
                const u32 sub = word >> (2 * j);
                const u32x4 x0 = *reinterpret_cast<LDSQ const u32x4 *>(lds + LY::TAB + 256 + ((sub & 3) << 4));
                const u32x4 x1 = *reinterpret_cast<LDSQ const u32x4 *>(lds + LY::TAB + 256 + ((sub & 0xc) << 2));
                const u32x4 x2 = *reinterpret_cast<LDSQ const u32x4 *>(lds + LY::TAB + 256 + (sub & 0x30));
                const u32x4 x3 = *reinterpret_cast<LDSQ const u32x4 *>(lds + LY::TAB + 256 + ((sub & 0xc0) >> 2));
                roll(x0);
                roll(x1);
                roll(x2);
                roll(x3);
            }
            for (; j < nb; ++j)
                roll(*reinterpret_cast<LDSQ const u32x4 *>(lds + LY::TAB + 256 + (((word >> (2 * j)) & 3) << 4)));
        }
        {  // block 0's words come from the preloaded four (k <= 33; beyond, from memory)
            const u32 wi = ((u32)k - 1) >> 4;
            in_lo = first_word(wi);
            in_hi = first_word(wi + 1);
            out_lo = pw.x;
            out_hi = pw.y;
            if (W > 16) {
                in_h2 = first_word(wi + 2);
                out_h2 = pw.z;
            }
        }
    }

    // One first block and ONE steady block variant (per-lane bound check, bounded staging store).  The uniform-wave and
    // unguarded variants each saved an instruction or two per step but made the kernel 65 KB of code against a 64 KB
    // instruction cache shared by two CUs: with the single variant (22 KB) the whole kernel runs 4 % faster.
    __device__ __forceinline__ void run(u32 nk_max) {
        begin();
        const u32 col8 = (u32)(lane & 31) * 8u;
        const bool up = lane < 32;
        nku = nk_max;
        block<true, false>(0);  // a lane stages at most W <= CAP tuples in its first block
        for (u32 i0 = W; i0 < nk_max; i0 += W) block<false, !DIRECT>(i0);
        if (PAIR) cnt = ((up ? slot : slim + col8 - slot)) / (u32)(LY::ROW * 8);  // col8 < ROW*8: the quotient is the row count
        else cnt = (slot - (u32)lane * 8u) / (u32)(LY::ROW * 8);
    }
};

// LDS -> HBM copy-out of one unit's staged tuples in read order (shared by the fast kernels).
// Owner of output t = the lane whose run [excl, excl + cnt) holds t: one head bit per non-empty lane (s_heads, one 64-bit word per
// row of 64 outputs), population count up to the output's bit -> rank among the non-empty lanes -> that lane's table entry
// {A, S}: the byte offset of output t inside the staged hashes is A + S t (S = one row up or -- the high lane of a paired column --
// down).  Per row of 64 outputs: two mbcnt on the (scalar) head word, one 8-byte table read, one 24-bit multiply-add, the two
// staged reads and the two stores; the first version (head word -> rank -> lane -> its first output -> row and column arithmetic
// per output, a quarter-rate 32-bit multiply among it, a bound check on every output) was 13 % of the minimizer kernel's time.
// PAIR: the staging of PLds (row = e for the low lane of a column, R-1-e for the high one); CAP is then the last head word.
template <class LY, bool POS16, int CAP, int U = 4, bool PAIR = false>
__device__ __forceinline__ void fast_copyout(char *lds, int lane, u32 cnt, u32 excl, u32 T, u64 base, const KArgs &a) {
    u64 *s_heads = reinterpret_cast<u64 *>(lds + LY::HEADS);
    // the table: {A, S} (8 bytes, PLds::CTAB) for paired columns; private columns all step upwards, S is a constant and the entry is A
    // alone (4 bytes, in the 256 bytes that held the exclusive offsets: SynLds has no room for more at eight waves per CU)
    u64 *s_tab64 = nullptr;
    if constexpr (PAIR) s_tab64 = reinterpret_cast<u64 *>(lds + LY::CTAB);
    u32 *s_tab32 = reinterpret_cast<u32 *>(lds + LY::EXCL);
    const u64 nzmask = __builtin_amdgcn_ballot_w64(cnt > 0);
    if (lane <= CAP) s_heads[lane] = 0;
    wave_sync_lds();
    if (cnt > 0) {
        constexpr int RB = LY::ROW * 8;  // bytes from one row of staged hashes to the next
        int A, S;
        if (PAIR) {
            const int col8 = (lane & 31) * 8;
            S = lane < 32 ? RB : -RB;
            A = lane < 32 ? col8 - (int)excl * RB : col8 + ((int)(BSK_PAIR_ROWS - 1) + (int)excl) * RB;
        } else {
            S = RB;
            A = lane * 8 - (int)excl * RB;
        }
        const u32 rk = __builtin_amdgcn_mbcnt_hi((u32)(nzmask >> 32), __builtin_amdgcn_mbcnt_lo((u32)nzmask, 0));
        if (PAIR) s_tab64[rk] = ((u64)(u32)S << 32) | (u32)A;
        else s_tab32[rk] = (u32)A;
        atomicOr(&s_heads[excl >> 6], 1ULL << (excl & 63));
    }
    wave_sync_lds();
    u32 heads_before = 0;  // wave-uniform: the head words are made scalar, their population count is SALU work
    const char *sh = lds + LY::SH, *sp = lds + LY::SP;
    // CHECK: the trip may reach beyond output T (the unit's last trip); every other trip runs without the per-output bound.
    // U rows per trip: their dependent LDS reads (head word -> table entry -> staged tuple) are issued together.
    auto trip = [&](u32 t0, auto check) {
        constexpr bool CHECK = decltype(check)::value;
        u32 rank[U], so[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const u32 c = (t0 >> 6) + j;
            const u64 Mv = s_heads[c < (u32)CAP ? c : (u32)CAP];  // word CAP is never set: rows beyond the unit see no heads
            const u32 mlo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)Mv), mhi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(Mv >> 32));
            const u64 M = ((u64)mhi << 32) | mlo;
            // heads at outputs <= this lane's = bits 1 .. lane of M (mbcnt of M >> 1 counts exactly those) + bit 0
            const u64 M1 = M >> 1;
            const u32 sbase = heads_before + (mlo & 1u) - 1u;  // scalar
            rank[j] = __builtin_amdgcn_mbcnt_hi((u32)(M1 >> 32), __builtin_amdgcn_mbcnt_lo((u32)M1, sbase));
            heads_before += (u32)__builtin_popcountll(M);
        }
        u64 ent[U];
#pragma unroll
        for (int j = 0; j < U; ++j) ent[j] = PAIR ? s_tab64[rank[j] & 63u] : (((u64)(u32)(LY::ROW * 8) << 32) | s_tab32[rank[j] & 63u]);
        u64 hv[U];
        u32 pv[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const u32 t = t0 + 64 * j + lane;
            const int off = __mul24((int)t, (int)(u32)(ent[j] >> 32)) + (int)(u32)ent[j];  // v_mad_i32_i24
            so[j] = (!CHECK || t < T) ? (u32)off : 0u;
            hv[j] = *reinterpret_cast<const u64 *>(sh + so[j]);
            if (POS16) pv[j] = (u32)(int)*reinterpret_cast<const short *>(sp + (so[j] >> 2));  // sign-extending read: the strand bit lands in bit 31
            else pv[j] = *reinterpret_cast<const u32 *>(sp + (so[j] >> 1));
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const u32 t = t0 + 64 * j + lane;
#ifdef BSK_FAST_NOSTORE  // dev: the copy-out's LDS chains without its stores (keeps the values alive)
            asm volatile("" ::"v"(hv[j]), "v"(pv[j]));
#else
            if (!CHECK || t < T) {
                __builtin_nontemporal_store(hv[j], &a.hash[base + t]);  // write-once output: non-temporal, so that the tuples streaming out do not push the sequences' lines out of the L2
                __builtin_nontemporal_store(POS16 ? (pv[j] & 0x80007fffu) : pv[j], &a.pos[base + t]);
            }
#endif
        }
    };
    u32 t0 = 0;
    for (; t0 + 64 * U <= T; t0 += 64 * U) trip(t0, std::false_type{});
    if (t0 < T) trip(t0, std::true_type{});
    wave_sync_lds();
}

template <int W, int CAP, bool POS16>
__global__ __launch_bounds__(64, (W >= 16 ? 2 : 1)) void k_minimizer_fast(KArgs a) {  // W >= 16: capped at 256 VGPRs (two waves per SIMD)
    constexpr bool PAIR = POS16;  // paired staging columns (8 waves per CU); 32-bit positions keep the private columns
    typedef typename MinLds<PAIR, CAP, POS16, BSK_PAIR_ROWS>::type LY;
    __shared__ __attribute__((aligned(16))) char lds[LY::TOTAL];
    LDSQ char *const ldsq = (LDSQ char *)lds;
    const int lane = lane_id();
    build_xtab(reinterpret_cast<uint4 *>(lds + LY::TAB), a.k, lane);
    if (PAIR && lane < 16) {  // two-base warm-up table: entry (c0 | c1 << 2), c0 entering first, nothing leaving
        const unsigned c0 = (unsigned)lane & 3u, c1 = (unsigned)lane >> 2;
        const u64 f = rol64(seed_fwd_code(c0), 1) ^ seed_fwd_code(c1);
        const u64 r = ror64(rol64(seed_rev_code(c0), (unsigned)(a.k - 1)), 1) ^ rol64(seed_rev_code(c1), (unsigned)(a.k - 1));
        reinterpret_cast<uint4 *>(lds + LY::TAB2)[lane] = make_uint4((u32)f, (u32)(f >> 32), (u32)r, (u32)(r >> 32));
    }
    __syncthreads();
    const u64 slab = (u64)64 * CAP;
    u64 d_next = 0;
    u32x4 pw_next = {0, 0, 0, 0};
    bool pre = false;
    // work distribution: a wave takes 8 consecutive units per ticket (one atomic per 512 reads)
    const u32 tku = a.tk ? a.tk : 8u;  // (KArgs::tk: fewer for small batches)
    for (u32 unit = next_ticket(a.ticket, lane) * tku, uend = unit + tku; unit < a.nunits; ++unit, ({
             if (unit == uend) {
                 unit = next_ticket(a.ticket, lane) * tku;
                 uend = unit + tku;
             }
         })) {
        const u64 r = (u64)unit * 64 + lane;
        // The descriptor and the first four words of a unit are loaded one unit ahead (within a ticket): a load issued at
        // the unit's start returns only after the previous unit's copy-out stores have drained (loads and stores share the
        // in-order vmcnt) and then costs two dependent memory latencies before the first base can be hashed.
        u64 d;
        u32x4 pw;
        if (pre) {
            d = d_next;
            pw = pw_next;
        } else {
            d = r < a.n ? a.desc[r] : 0;
            pw = *reinterpret_cast<const GLBQ u32x4_u *>((size_t)(a.words + (d >> 24)));
        }
        const u64 off = d >> 24, L = desc_len(a, d);
        const bool nxt = unit + 1 != uend && unit + 1 < a.nunits;
        if (nxt) d_next = r + 64 < a.n ? a.desc[r + 64] : 0;
        const bool ok = r < a.n && L >= (u64)a.circ_ext && (L - (u64)a.circ_ext) + 1 >= (u64)a.k + (u64)W;
        const u32 nk = ok ? (u32)(L - a.k + 1) : 0u;
        const u32 nk_max = wave_max_u32(nk);
        u32 cnt = 0, tie = 0;
        if (nk_max) {
            FastMin<W, CAP, POS16, false, PAIR> fm;
            fm.w = a.words + off;
            fm.pw = pw;
            fm.lds = ldsq;
            fm.k = a.k;
            fm.lane = lane;
            fm.nk = nk;
            fm.run(nk_max);
            cnt = fm.cnt;
            tie = fm.tie;
        }
        if (nxt) pw_next = *reinterpret_cast<const GLBQ u32x4_u *>((size_t)(a.words + (d_next >> 24)));  // ahead of the copy-out stores
        pre = nxt;
        // ---- unit epilogue: LDS -> HBM in read order into the unit's slab ----
        const u32 incl = wave_incl_scan_u32(cnt, lane);
        const u32 excl = incl - cnt;
        const u32 T = wave_bcast_u32(incl, 63);
        // PAIR: a column is full when its two lanes' counts reach R together (its last free row takes the unselected candidates)
        const u32 cnt_pair = PAIR ? cnt + (u32)__builtin_amdgcn_ds_bpermute((lane ^ 32) * 4, (int)cnt) : 0u;
        const bool any_over = PAIR ? __builtin_amdgcn_ballot_w64(cnt_pair >= (u32)BSK_PAIR_ROWS) != 0 : __builtin_amdgcn_ballot_w64(cnt > (u32)CAP) != 0;
        u64 base = (u64)unit * slab;
        if (!any_over) {
            // copy-out rows in flight: what keeps the kernel within 256 VGPRs (two waves per SIMD)
#ifndef BSK_FAST_NOCOPYOUT
            if (PAIR) fast_copyout<LY, POS16, LY::NHEADS - 1, (W <= 11 ? 4 : W <= 15 ? 2 : 1), true>(lds, lane, cnt, excl, T, base, a);
            else fast_copyout<LY, POS16, CAP>(lds, lane, cnt, excl, T, base, a);
#endif  // (dev: timing without the copy-out)
        } else {
            // rare (0.6 % of units at k=21 w=11 CAP=32): a lane selected more than CAP tuples.  The unit may not
            // fit its slab: take T tuples from the overflow region and recompute, storing straight to HBM.
            u64 ob = 0;
            if (lane == 0) ob = atomicAdd(a.total + 1, (u64)T);
            ob = wave_bcast_u64(ob, 0);
            if (ob + T <= a.ovf_cap) {
                base = a.ovf_base + ob;
                FastMin<W, CAP, POS16, true> fm;
                fm.w = a.words + off;
                fm.pw = pw;
                fm.lds = ldsq;
                fm.k = a.k;
                fm.lane = lane;
                fm.nk = nk;
                fm.ghash = a.hash;
                fm.gpos = a.pos;
                fm.gbase = base + excl;
                fm.run(nk_max);
            } else {
                cnt = 0;  // result buffers too small: flagged, the host re-runs with a larger overflow region
                if (lane == 0) atomicOr(&a.ticket[1], 1u);
            }
        }
        if (r < a.n) {
            a.refs[r] = ((base + excl) << 24) | cnt;
            u8 sbyte = ok ? BSK_ST_OK : BSK_ST_SHORT;
            if (tie && ok) sbyte |= BSK_ST_FIRST_WINDOW_TIE;
            if (ok && a.rflags) sbyte |= a.rflags[r];
            a.status[r] = sbyte;
        }
    }
}

// Flush staged tuples to the per-sequence slabs.  Regular rounds move only whole groups of 16 tuples per lane
// (a full, aligned 128-byte line of hashes and 64 bytes of positions: no partial-line writes; measured 2.2x
// faster than flushing every tuple of every round); the final round moves what is left.
// STRAND16: staged positions carry the strand in bit 15 (DNA kernels); it moves to bit 31 on the way out.
// GL = log2 of the tuples per flushed group.  Groups of 16 are whole 128-byte lines of hashes but keep up to 15 left-over rows
// per lane staged; groups of 8 (aligned 64-byte pieces) need 8 rows less -- two more waves per CU, protein minimizer w=5
// 366 -> 424 G residues/s -- but every half-line write is a read-modify-write in HBM (traffic 1.3x -> 2.2x the algorithmic
// bytes).  The kernels therefore keep GL = 4 and flush more often instead (fewer new rows between two flushes).
// RING: the lane's rows are a ring of RING rows starting at row `head` (no left-over moves).
//
// The regular flush round of the per-read-slab kernels: every lane sends its whole groups of 1 << GL staged tuples (a group = one
// 128-byte line of hashes) to its own slab.  A work unit is ONE group, so the owner of an output needs no bitmap: unit u = t >> GL is
// entry u of a 4-byte table {owner lane : 6, first ring row of the group : 6, its tuple index inside the owner's slab : 20} that the
// owners fill (one or two entries each); an output then costs one broadcast table read, the ring wrap, one 24-bit multiply-add each
// for the staged slot and the destination, the two staged reads and the two stores, and the rows of a round (three or four) go as
// one batch.  The first version (flush_rows) -- head bitmap built with LDS atomics, rank by population count, owner, its offset / destination / ring
// head from three more arrays, a 32-bit multiply, one row at a time behind two hand-offs -- ran this round every ten steps of the
// protein minimizer and was 47 % of that kernel (33.2 ms with it, 17.4 ms with the flushes compiled out).
// lost: where a lane whose slab is too small says so (k_minimizer_pkd sends that read to the exact machine's list); null: the call is sized again
template <class LY, bool STRAND16, int GL, int RING>
__device__ __forceinline__ void flush_groups(char *lds, int lane, u32 cnt, u32 done, u64 slab_read, u64 ubase, const KArgs &a, u32 head, u32 *lost = nullptr) {
    constexpr u32 GM = (1u << GL) - 1u;
    u32 *s_tab = reinterpret_cast<u32 *>(lds + LY::DST);  // 128 entries: a lane holds fewer than 2 << GL + ... staged tuples (RING <= 2 << GL + GM)
    static_assert(RING <= 64 && ((RING - 1) >> GL) <= 2, "a lane flushes at most two groups per round");
    const u32 units = cnt >> GL;
    const u32 incl = wave_incl_scan_u32(units, lane);
    const u32 excl = incl - units;
    const u32 U = wave_bcast_u32(incl, 63);
    if (U == 0) return;
    const bool fits = (u64)done + ((u64)units << GL) <= slab_read && (u64)done + ((u64)units << GL) < (1u << 20);
    if (lost) *lost |= fits ? 0u : 1u;
    else if (__builtin_amdgcn_ballot_w64(!fits) && lane == 0) atomicOr(&a.ticket[1], 1u);  // slab too small: host falls back
    if (units > 0) {
        u32 row0 = head, doff = fits ? done : 0xfffffu;
        s_tab[excl] = ((u32)lane << 26) | (row0 << 20) | doff;
        if (units > 1) {
            row0 += (1u << GL);
            row0 = row0 >= (u32)RING ? row0 - (u32)RING : row0;
            s_tab[excl + 1] = ((u32)lane << 26) | (row0 << 20) | (fits ? done + (1u << GL) : 0xfffffu);
        }
    }
    wave_sync_lds();
    const u32 T = U << GL;
    const u32 slab24 = (u32)slab_read;  // < 2^24 (the kernels' callers bound the sequence length)
    // A lane moves TWO consecutive tuples (same group, hence same owner): one table read, one 16-byte store of hashes and one
    // 8-byte store of positions per pair.  UR rows of 128 outputs per trip: their table and staged reads are issued together.
    constexpr int UR = 2;
    for (u32 t0 = 0; t0 < T; t0 += 128 * UR) {
        u32 ent[UR];
#pragma unroll
        for (int j = 0; j < UR; ++j) {
            const u32 t = t0 + 128 * j + 2 * (u32)lane;
            ent[j] = s_tab[(t < T ? t : T - 1) >> GL];
        }
        u64 h0[UR], h1[UR];
        u32 p0[UR], p1[UR], di[UR];
        bool ok[UR];
#pragma unroll
        for (int j = 0; j < UR; ++j) {
            const u32 t = t0 + 128 * j + 2 * (u32)lane;
            const u32 owner = ent[j] >> 26, sub = t & GM, doff = ent[j] & 0xfffffu;
            u32 r0 = ((ent[j] >> 20) & 63u) + sub, r1 = r0 + 1;
            r0 = r0 < r0 - (u32)RING ? r0 : r0 - (u32)RING;  // ring wrap: r - RING wraps to a huge value when r < RING
            r1 = r1 < r1 - (u32)RING ? r1 : r1 - (u32)RING;
            const u32 s0 = __umul24(r0, (u32)LY::ROW) + owner, s1 = __umul24(r1, (u32)LY::ROW) + owner;
            ok[j] = t < T && doff != 0xfffffu;
            di[j] = __umul24(owner, slab24) + doff + sub;
            h0[j] = *reinterpret_cast<const u64 *>(lds + LY::SH + s0 * 8);
            h1[j] = *reinterpret_cast<const u64 *>(lds + LY::SH + s1 * 8);
            p0[j] = *reinterpret_cast<const u16 *>(lds + LY::SP + s0 * 2);
            p1[j] = *reinterpret_cast<const u16 *>(lds + LY::SP + s1 * 2);
        }
#pragma unroll
        for (int j = 0; j < UR; ++j) {
#ifdef BSK_FLUSH_NOSTORE  // dev: the flush's work without its stores
            asm volatile("" ::"v"(h0[j]), "v"(h1[j]), "v"(p0[j]), "v"(p1[j]), "v"(di[j]), "v"((u32)ok[j]));
#else
            if (ok[j]) {
                const u32 q0 = STRAND16 ? (p0[j] & 0x7fffu) | ((p0[j] & 0x8000u) << 16) : p0[j];
                const u32 q1 = STRAND16 ? (p1[j] & 0x7fffu) | ((p1[j] & 0x8000u) << 16) : p1[j];
                // write-once output: non-temporal, so that the tuples streaming out do not push the sequences' lines out of the L2
                nt_store_u64x2(a.hash + ubase + di[j], h0[j], h1[j]);
                __builtin_nontemporal_store(((u64)q1 << 32) | q0, reinterpret_cast<u64 *>(a.pos + ubase + di[j]));
            }
#endif
        }
    }
    wave_sync_lds();
}

// The last round of a sequence: whole groups as above, then every lane's left-over (fewer than a group) as ONE partial group -- the
// same table with the lane's count beside the entry, outputs beyond the count masked.  (Sixteen rows of 64 outputs instead of the
// seven or eight dense ones of the first version, but four at a time and without its chain of five dependent LDS reads per row.)
template <class LY, bool STRAND16, int GL, int RING>
__device__ __forceinline__ void flush_last(char *lds, int lane, u32 cnt, u32 done, u64 slab_read, u64 ubase, const KArgs &a, u32 head, u32 *lost = nullptr) {
    constexpr u32 GM = (1u << GL) - 1u;
    flush_groups<LY, STRAND16, GL, RING>(lds, lane, cnt, done, slab_read, ubase, a, head, lost);
    const u32 g = cnt & ~GM;
    head += g;
    head = head >= (u32)RING ? head - (u32)RING : head;
    head = head >= (u32)RING ? head - (u32)RING : head;
    done += g;
    const u32 c = cnt - g;  // < 1 << GL
    u32 *s_tab = reinterpret_cast<u32 *>(lds + LY::DST);  // 64 entries of {owner, row, destination} + 64 counts
    const u32 unit = c ? 1u : 0u;
    const u32 incl = wave_incl_scan_u32(unit, lane);
    const u32 U = wave_bcast_u32(incl, 63);
    if (U == 0) return;
    const bool fits = (u64)done + c <= slab_read && (u64)done + c < (1u << 20);
    if (lost) *lost |= fits ? 0u : 1u;
    else if (__builtin_amdgcn_ballot_w64(!fits) && lane == 0) atomicOr(&a.ticket[1], 1u);
    if (c) {
        s_tab[incl - 1] = ((u32)lane << 26) | (head << 20) | (fits ? done : 0xfffffu);
        s_tab[64 + incl - 1] = c;
    }
    wave_sync_lds();
    const u32 T = U << GL;
    const u32 slab24 = (u32)slab_read;
    constexpr int UR = 4;
    for (u32 t0 = 0; t0 < T; t0 += 64 * UR) {
        u32 ent[UR], cn[UR];
#pragma unroll
        for (int j = 0; j < UR; ++j) {
            const u32 t = t0 + 64 * j + lane;
            const u32 u = (t < T ? t : T - 1) >> GL;
            ent[j] = s_tab[u];
            cn[j] = s_tab[64 + u];
        }
        u64 hv[UR];
        u32 pv[UR], di[UR];
        bool ok[UR];
#pragma unroll
        for (int j = 0; j < UR; ++j) {
            const u32 t = t0 + 64 * j + lane;
            const u32 owner = ent[j] >> 26, sub = t & GM, doff = ent[j] & 0xfffffu;
            u32 row = ((ent[j] >> 20) & 63u) + sub;
            row = row < row - (u32)RING ? row : row - (u32)RING;
            ok[j] = t < T && sub < cn[j] && doff != 0xfffffu;
            const u32 sl = ok[j] ? __umul24(row, (u32)LY::ROW) + owner : 0u;
            di[j] = __umul24(owner, slab24) + doff + sub;
            hv[j] = *reinterpret_cast<const u64 *>(lds + LY::SH + sl * 8);
            pv[j] = *reinterpret_cast<const u16 *>(lds + LY::SP + sl * 2);
        }
#pragma unroll
        for (int j = 0; j < UR; ++j) {
            if (ok[j]) {
                const u32 pp = STRAND16 ? (pv[j] & 0x7fffu) | ((pv[j] & 0x8000u) << 16) : pv[j];
                __builtin_nontemporal_store(hv[j], &a.hash[ubase + di[j]]);
                __builtin_nontemporal_store(pp, &a.pos[ubase + di[j]]);
            }
        }
    }
    wave_sync_lds();
}

// ---------------------------------------------------------------------------------------
// Dense minimizers (small w): a 150-bp read at w = 5 selects ~43 positions, more than the 32 k_minimizer_fast stages per
// read, so almost every unit took its recompute-and-store-directly path (w = 5: 250 Gbases/s against 620 at w = 10).
// Here every read owns a slab of `slab_read` tuples (as the protein minimizer does) and the wavefront flushes its staging
// every NB blocks in whole groups of 16 tuples per read (flush_groups); fewer than 16 stay staged until the next round.
// ---------------------------------------------------------------------------------------
template <int W>
struct DenseCfg {
    // blocks per flush round: <= 13 steps, so that 15 left-over rows + the new ones + the scribble row leave 8 waves per CU.
    // (<= 24 steps: 5-6 waves; groups of 8 tuples with <= 20 steps: 8 waves and as fast, but half-line writes double the traffic.)
    static constexpr int NB = (13 / W) > 0 ? (13 / W) : 1;
#ifndef BSK_DENSE_GL
#define BSK_DENSE_GL 4
#endif
    static constexpr int GL = BSK_DENSE_GL;
    static constexpr int G = 1 << GL;
    static constexpr int CAP = NB * W + G - 1;              // rows = CAP + 1: the left-over of a group + NB*W new + the scribble row
};

// LIST: the READS k_minimizer_pk / k_minimizer_ring listed (list_append, kernels_generic.hpp: one segment of a.rlist per workgroup of the main launch) -- reads in
// which two equal 27-bit keys met in a min operation.  Real ties are most of them on real data: low-complexity reads (poly-A / poly-G
// tails, repeats), and a homopolymer selects EVERY position -- which is this kernel's case (per-read slabs, mid-read flushes), not
// k_minimizer_fast's, whose staging columns such reads overflow.  64 listed reads per wavefront; every read gets a slab of
// a.slab_read tuples (the caller passes the largest possible count) in the overflow region.
// ASC (round 5): the side launch of a mixed batch -- the reads a.subset names, from their ASCII bytes (a.ascii, a.aoff), every read a slab of
// a.slab_read tuples (the caller passes one per window: nothing to outgrow) in [a.out_base, a.cap).  The general per-lane ASCII kernel it
// replaces keeps its window in global memory and ran 1 % of a batch's reads in a third of the batch's time.
template <int W, bool LIST = false, bool ASC = false>
__global__ __launch_bounds__(64) void k_minimizer_dense(KArgs a) {
    constexpr int CAP = DenseCfg<W>::CAP, NB = DenseCfg<W>::NB, GL = DenseCfg<W>::GL, G = DenseCfg<W>::G;
    typedef typename std::conditional<ASC, FLdsA<CAP>, FLds<CAP, true>>::type LY;
    __shared__ __attribute__((aligned(16))) char lds[LY::TOTAL];
    LDSQ char *const ldsq = (LDSQ char *)lds;
    const int lane = lane_id();
    // LIST: this workgroup's segment of the list (list_append, kernels_generic.hpp)
    const u32 lseg = LIST ? a.fixcap / a.list_grid : 0u;
    if (LIST) {  // (nothing listed in any of this workgroup's segments: leave before the table is built)
        u32 any = 0;
        for (u32 sg = blockIdx.x; sg < a.list_grid; sg += gridDim.x) any |= a.rlist[sg];
        if (!any) return;
    }
    if constexpr (ASC) build_bytetabs(reinterpret_cast<uint4 *>(lds + FLdsA<CAP>::TIN), reinterpret_cast<uint4 *>(lds + FLdsA<CAP>::TOUT), a.k, lane);
    else build_xtab(reinterpret_cast<uint4 *>(lds + LY::TAB), a.k, lane);
    __syncthreads();
    const u64 slab_read = a.slab_read;
    for (u32 sg = LIST ? blockIdx.x : 0u; sg < (LIST ? a.list_grid : 1u); sg += LIST ? gridDim.x : 1u) {
    const u32 nlist = LIST ? (a.rlist[sg] < lseg ? a.rlist[sg] : lseg) : 0u;
    const u32 *const mylist = LIST ? a.rlist + a.list_grid + sg * lseg : nullptr;
    const u32 TK = ASC ? 1u : (a.tk ? a.tk : 4u);  // units per ticket (the side launch's few units are latency: one per wavefront; KArgs::tk: fewer for small batches)
    for (u32 unit = LIST ? 0u : next_ticket(a.ticket, lane) * TK, uend = unit + TK; LIST ? unit < (nlist + 63u) / 64u : unit < a.nunits; ++unit, ({
             if (!LIST && unit == uend) {
                 unit = next_ticket(a.ticket, lane) * TK;
                 uend = unit + TK;
             }
         })) {
        u64 r = (u64)unit * 64 + lane;
        if (LIST) r = r < nlist ? (u64)mylist[r] : ~0ULL;
        if (ASC) r = r < a.nsub ? (u64)a.subset[r] : ~0ULL;
        u64 off = 0, L = 0, ro = r;
        if (r < a.n) {
            if constexpr (ASC) {
                ascii_span(a, r, off, L);  // (byte offset and length of the read's ASCII)
            } else {
                const u64 d = a.desc[r];
                off = d >> 24;
                L = desc_len(a, d);
                ro = out_index(a, r, d);  // (length-binned batches: the read's own place in its chunk)
            }
        }
        const bool ok = r < a.n && L >= (u64)a.circ_ext && (L - (u64)a.circ_ext) + 1 >= (u64)a.k + (u64)W;
        const u32 nk = ok ? (u32)(L - a.k + 1) : 0u;
        const u32 nk_max = wave_max_u32(nk);
        u64 ubase = (u64)unit * 64 * slab_read;
        bool room = true;
        if (ASC) {  // slabs of the side launch's own region
            ubase += a.out_base;
            room = ubase + 64 * slab_read <= a.cap;
            if (!room && lane == 0) atomicOr(&a.ticket[1], 1u);
            if (unit == a.nunits - 1 && lane == 0) *a.total = ubase + 64 * slab_read;
        }
        if (LIST) {  // 64 slabs from the overflow region
            u64 ob = 0;
            if (lane == 0) ob = atomicAdd(a.total + 1, 64 * slab_read);
            ob = wave_bcast_u64(ob, 0);
            room = ob + 64 * slab_read <= a.ovf_cap;
            if (!room && lane == 0) atomicOr(&a.ticket[1], 1u);  // the host re-runs with a larger overflow region
            ubase = a.ovf_base + ob;
        }
        u32 done = 0, tie = 0;
        if (nk_max && room) {
            FastMin<W, CAP, true, false, false, true, BSK_PAIR_ROWS, 0, ASC> fm;
            fm.w = a.words + off;
            fm.ab = ASC ? a.ascii + off : nullptr;
            fm.in_lo = fm.in_hi = fm.out_lo = fm.out_hi = 0;
            fm.send = (u32)((CAP + 1) * LY::ROW + lane) * 8u;
            fm.lds = ldsq;
            fm.k = a.k;
            fm.lane = lane;
            fm.nk = nk;
            fm.fl = fm.fh_ = fm.rl = fm.rh_ = 0;
            fm.prev = 0xffffffffu;
            fm.tie = 0;
            fm.nku = 0xffffffffu;
            fm.slot = (u32)lane * 8u;
            if constexpr (ASC) {  // warm-up from the bytes: tin rows alone (nothing leaves)
                int t = 0;
                for (; t + 4 <= a.k - 1; t += 4) {
                    const u32 b0 = fm.ab[t], b1 = fm.ab[t + 1], b2 = fm.ab[t + 2], b3 = fm.ab[t + 3];
                    const u32x4 x0 = *reinterpret_cast<LDSQ const u32x4 *>(ldsq + LY::TIN + (b0 << 4)), x1 = *reinterpret_cast<LDSQ const u32x4 *>(ldsq + LY::TIN + (b1 << 4));
                    const u32x4 x2 = *reinterpret_cast<LDSQ const u32x4 *>(ldsq + LY::TIN + (b2 << 4)), x3 = *reinterpret_cast<LDSQ const u32x4 *>(ldsq + LY::TIN + (b3 << 4));
                    fm.roll(x0);
                    fm.roll(x1);
                    fm.roll(x2);
                    fm.roll(x3);
                }
                for (; t < a.k - 1; ++t) fm.roll(*reinterpret_cast<LDSQ const u32x4 *>(ldsq + LY::TIN + ((u32)fm.ab[t] << 4)));
            } else {
            for (int t0 = 0; t0 < a.k - 1; t0 += 16) {  // warm-up: bases 0..k-2 enter, nothing leaves
                const u32 word = fm.w[t0 >> 4];
                const int nb = (a.k - 1 - t0) < 16 ? (a.k - 1 - t0) : 16;
                int j = 0;
                for (; j + 4 <= nb; j += 4) {  // four table rows in flight
                    const u32 sub = word >> (2 * j);
                    const u32x4 x0 = *reinterpret_cast<LDSQ const u32x4 *>(ldsq + LY::TAB + 256 + ((sub & 3) << 4));
                    const u32x4 x1 = *reinterpret_cast<LDSQ const u32x4 *>(ldsq + LY::TAB + 256 + ((sub & 0xc) << 2));
                    const u32x4 x2 = *reinterpret_cast<LDSQ const u32x4 *>(ldsq + LY::TAB + 256 + (sub & 0x30));
                    const u32x4 x3 = *reinterpret_cast<LDSQ const u32x4 *>(ldsq + LY::TAB + 256 + ((sub & 0xc0) >> 2));
                    fm.roll(x0);
                    fm.roll(x1);
                    fm.roll(x2);
                    fm.roll(x3);
                }
                for (; j < nb; ++j)
                    fm.roll(*reinterpret_cast<LDSQ const u32x4 *>(ldsq + LY::TAB + 256 + (((word >> (2 * j)) & 3) << 4)));
            }
            fm.load_block_words(0);
            }
            int inround = 0;
            u32 head = 0;
            for (u32 i0 = 0; i0 < nk_max; i0 += W) {
                if (i0 == 0) fm.template block<true, false>(0);  // one first, one steady variant: see FastMin::run
                else fm.template block<false, false>(i0);
                const bool last = i0 + W >= nk_max;
                if (++inround == NB || last) {
                    inround = 0;
#ifndef BSK_DENSE_LATE_WAIT
                    // the next block's words (requested a block of hashing ago) are waited for HERE, before the flush's stores are issued:
                    // vmcnt is one in-order counter, and a wait behind the stores -- where the compiler puts it, at the words' first
                    // use -- waits for every one of them to reach memory
                    asm volatile("" : "+v"(fm.in_lo), "+v"(fm.in_hi), "+v"(fm.out_lo), "+v"(fm.out_hi));
#endif
                    // the lane's rows are a ring starting at `head` (moving the left-over down after every flush cost more than the wrap test)
                    const u32 wrow = (fm.slot - (u32)lane * 8u) / (u32)(LY::ROW * 8);
                    const u32 cnt = wrow >= head ? wrow - head : wrow + (u32)(CAP + 1) - head;  // staged: left-over < 16 + NB*W new
                    if (last) flush_last<LY, true, GL, CAP + 1>(lds, lane, cnt, done, slab_read, ubase, a, head);
                    else flush_groups<LY, true, GL, CAP + 1>(lds, lane, cnt, done, slab_read, ubase, a, head);
                    const u32 nfl = last ? cnt : (cnt & ~(u32)(G - 1));
                    head += nfl;
                    head = head >= (u32)(CAP + 1) ? head - (u32)(CAP + 1) : head;
                    done += nfl;
                }
            }
            tie = fm.tie;
        }
        if (r < a.n) {
            a.refs[ro] = ((ubase + (u64)lane * slab_read) << 24) | done;
            u8 sbyte = ok ? BSK_ST_OK : BSK_ST_SHORT;
            if (tie && ok) sbyte |= BSK_ST_FIRST_WINDOW_TIE;
            if (ok && a.rflags) sbyte |= a.rflags[r];
            a.status[ro] = sbyte;
        }
    }
    }
}

// ---------------------------------------------------------------------------------------
// ntHash stream (kind BSK_NTHASH), 2-bit input: value i of read r -> hash[first(r) + i].
// Same rolling core as above, unrolled by 16 (one packed word of bases per block); the 16
// hashes of a block go through a 64x16 LDS tile (row = read) so that they leave as 128
// contiguous bytes per read, 16 bytes per lane and store.  For fixed-length batches the
// output offset of every read is a closed form, otherwise it comes from the look-back.
// ---------------------------------------------------------------------------------------
#define BSK_NT_FAST_WORDS 34  // reads of up to 32*16 = 512 bases (+2 words of look-ahead)
// MODE 0: ntHash forward strand, 1: canonical ntHash, 2: canonical 2-bit k-mer code (NextKmer, iterator.go:708-759, k <= 32),
// 3 (round 6): NextKmer's TWO-STRAND mode (iterator.go:713-723: the codes of the sequence, then the codes of its reverse complement) --
// a read's run is 2 nk values: value j < nk is the forward code at position j, value j >= nk the reverse-complement code of position
// 2 nk - 1 - j, i.e. the lane walks forward through the read and then BACK, rolling the reverse-complement code the other way (a base
// enters at the k-mer's front); the runs leave in whole lines exactly as in the other modes, 16 bytes per base (the general kernel this
// replaces ran 166 Gbases/s); 4: forward codes only (the tile pass of long sequences in two-strand mode, k_two_strand after it)
// CP ("compact runs", fixed-length batches of reads with >= 32 values; an EXPERIMENT -- measured and rejected, see stream_compact_ok in
// biosketch.hip -- instantiated only with make EXPERIMENTS=1): no padding at all.  Read l of a unit starts at
//   unit base + l * nk  (the unit's 64 nk values are one run; 64 nk is a multiple of 16, so units start on lines)
// and the 128-byte lines are written WHOLE all the same: the tile row is a ring of the last 32 values of a read; after block b a
// read's line b -- values [16 b - e, 16 b + 16 - e), e = the read's start modulo 16 -- leaves if it lies inside the read, and the
// lines that hold the tail of one read and the head of the next (one per read) leave at the end of the unit, assembled from the
// ring and from the heads, which every lane keeps in registers since block 0.  Padded runs cost 144 values written for 130
// (k = 21, 150 bases): 11 % of the write traffic of a write-bound kernel.
__host__ __device__ __forceinline__ unsigned pair_map2(int pairs);  // (kernels_more.hpp: the 2-bit code of a base's PairLetter in the batch's alphabet)
template <int MODE, bool CP = false>
__global__ __launch_bounds__(64) void k_nthash_fast(KArgs a) {
    constexpr int TL = CP ? 34 : 18;  // u64 per tile row (16 + 2 pad: 144-byte rows keep 16-byte alignment, 2-way conflicts at most; CP: a ring of 32 + 2)
    constexpr int NWL = BSK_NT_FAST_WORDS;          // packed words of a read staged in LDS
    constexpr int SW_OFF = 512 + 64 * TL * 8;
    static_assert(!CP || NWL * 64 * 4 >= 64 * 17 * 8, "the heads of a unit take the words' place");
    __shared__ __attribute__((aligned(16))) char lds[SW_OFF + NWL * 64 * 4];
    __shared__ u64 s_off[64];
    __shared__ u32 s_nk[64];
    LDSQ char *const lq = (LDSQ char *)lds;
    const int lane = lane_id();
    build_xtab(reinterpret_cast<uint4 *>(lds), a.k, lane);
    __syncthreads();
    const int k = a.k;
    __shared__ u64 s_base[8];
    u32 wa0[CP ? 16 : 1], wa1[CP ? 16 : 1];  // CP: LDS addresses of this lane's 16 values of an even / odd block
    if constexpr (CP) {
        const u32 e = ((u32)lane * (a.uniform_len - (u32)k + 1u)) & 15u;  // the read's start modulo 16 (reads of a unit are nk apart)
#pragma unroll
        for (int o = 0; o < 16; ++o) {
            wa0[o] = 512u + (u32)lane * (u32)(TL * 8) + (((u32)o + e) & 31u) * 8u;
            wa1[o] = 512u + (u32)lane * (u32)(TL * 8) + ((16u + (u32)o + e) & 31u) * 8u;
        }
    }
    for (;;) {
      const u32 u0 = next_ticket(a.ticket, lane) * 8u;
      if (u0 >= a.nunits) break;
      const u32 u1 = u0 + 8u < a.nunits ? u0 + 8u : a.nunits;
      if (!a.uniform_len) {
          // ragged batch: the output offsets of ALL units of this ticket are resolved (and published) before any of them
          // is processed -- a wave that published unit u only after processing units u0..u-1 serialised the whole grid
          // behind the look-back chain (measured: 200 Mbases of 277-base tiles took 210 ms instead of 2)
          // (the scan element of the look-back chain is the TICKET: one aggregate for its 8 units, published at once)
          u64 run = 0;
          for (u32 unit = u0; unit < u1; ++unit) {
              const u64 r = (u64)unit * 64 + lane;
              u64 L = 0;
              if (r < a.n) L = a.desc[r] & 0xffffffULL;
              const bool ok = r < a.n && L >= (u64)a.circ_ext && (L - (u64)a.circ_ext) >= (u64)k;
              const u32 pk = ok ? ((MODE == 3 ? 2u : 1u) * (u32)(L - k + 1) + 15u) & ~15u : 0u;
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
        const bool ok = r < a.n && L >= (u64)a.circ_ext && (L - (u64)a.circ_ext) >= (u64)k;  // iterator.go:619
        const u32 nkp = ok ? (u32)(L - k + 1) : 0u;      // k-mer positions of the read
        const u32 nk = MODE == 3 ? 2u * nkp : nkp;       // values of its run
        const u32 nkp_max = wave_max_u32(nkp);
        const u32 nk_max = MODE == 3 ? 2u * nkp_max : nkp_max;
        // every read's run starts on a 128-byte line and is padded to whole lines (16 values): each 16-step flush
        // then writes full, aligned lines (measured WRITE_SIZE 1.24x -> ~1.0x of the algorithmic bytes)
        const u32 pk = CP ? nk : (nk + 15u) & ~15u;
        const u64 incl = wave_incl_scan_u64((u64)pk, lane);
        const u64 T = wave_bcast_u64(incl, 63);
        const u64 base = CP ? (u64)unit * 64 * nk_max : a.uniform_len ? (u64)unit * 64 * ((nk_max + 15u) & ~15u) : s_base[unit - u0];
        const bool ovf = base + ((T + 15u) & ~(u64)15) > a.cap;  // (CP: the last line of a partial unit is written whole)
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
        // the 8 rows this lane serves during a flush: row = rr*8 + lane/8, two values at column (lane%8)*2
        u64 roff[8];
        u32 rnk[8];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            roff[rr] = s_off[rr * 8 + (lane >> 3)] + (u32)(lane & 7) * 2;
            rnk[rr] = s_nk[rr * 8 + (lane >> 3)];
        }
        u32 re[8];  // CP: the served row's start modulo 16 (its line b begins e values before its block b)
        if (CP) {
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                re[rr] = ((u32)(rr * 8 + (lane >> 3)) * nk_max) & 15u;
                roff[rr] -= re[rr];
            }
        }
        u64 hv[CP ? 16 : 1];  // CP: the read's first 16 values
        // the read's packed words go to LDS once ([word][lane]: conflict-free): the k-mer loop then has no global loads,
        // so its flush stores stay in flight across blocks (with loads in the loop the compiler's s_waitcnt vmcnt(0)
        // made every block wait for the previous flush to reach memory)
        const u32 *__restrict__ w = a.words + off;
        LDSQ u32 *const sw = reinterpret_cast<LDSQ u32 *>(lq + SW_OFF);
        const u32 nw_max = ((nkp_max + (u32)k - 1 + 15) >> 4) + 2;  // <= NWL (checked by the host)
        for (u32 j = 0; j < nw_max; ++j) sw[j * 64 + lane] = w[j];
        wave_sync_lds();
        u32 fl = 0, fh_ = 0, rl = 0, rh_ = 0;
        auto roll = [&](u32x4 x) {
            const u32 p = __builtin_amdgcn_alignbit(fl, fh_, 31), q = __builtin_amdgcn_alignbit(fh_, fl, 31);
            const u32 c = __builtin_amdgcn_alignbit(rh_, rl, 1), d = __builtin_amdgcn_alignbit(rl, rh_, 1);
            fl = p ^ x.x;
            fh_ = q ^ x.y;
            rl = c ^ x.z;
            rh_ = d ^ x.w;
        };
        u64 code = 0, rc = 0;  // MODE 2: forward / reverse-complement code, first base in the most significant pair
        // MODE 3: rc is the code of the reverse-COMPLEMENTED LETTERS (RevComInplace, iterator.go:719: PairLetter of the sequence's alphabet --
        // RNA leaves a T, Unlimit complements nothing), not the arithmetic complement the canonical mode compares with (iterator.go:740)
        const u32 map2 = MODE == 3 ? pair_map2(a.pairs) : 0x1Bu;
        auto cm = [&](u32 b) -> u32 { return MODE == 3 ? (map2 >> (2u * b)) & 3u : b ^ 3u; };
        const u64 cmask = k >= 32 ? ~0ULL : ((1ULL << (2 * k)) - 1ULL);
        const unsigned sh2 = 2u * (unsigned)(k - 1);
        for (int t0 = 0; t0 < k - 1; t0 += 16) {
            const u32 word = sw[(t0 >> 4) * 64 + lane];
            const int nb = (k - 1 - t0) < 16 ? (k - 1 - t0) : 16;
            for (int j = 0; j < nb; ++j) {
                const u32 b = (word >> (2 * j)) & 3;
                if (MODE >= 2) {
                    code = (code << 2) | b;
                    rc = (rc >> 2) | ((u64)cm(b) << sh2);
                } else {
                    roll(*reinterpret_cast<LDSQ const u32x4 *>(lq + 256 + (b << 4)));
                }
            }
        }
        LDSQ char *const myrow0 = lq + 512 + lane * (TL * 8);
        // one block of 16 values + its flush.  CP: the row is a ring of two blocks, PAR = which half this block's values go to -- shifted by
        // the read's start modulo 16 (wa0 / wa1), so that the ring columns [16 PAR, 16 PAR + 16) hold exactly line b of the row
        auto blockfn = [&](auto parc, const u32 i0) {
            constexpr int PAR = decltype(parc)::value;
            const u32 t0 = MODE == 3 ? (i0 + (u32)k - 1 < 16u * (NWL - 2) ? i0 + (u32)k - 1 : 16u * (NWL - 2)) : i0 + (u32)k - 1, p0 = MODE == 3 ? 0u : (i0 ? i0 - 1 : 0);
            u32 cb2 = 0;  // MODE 3: the 16 bases at positions P0 - 15 .. P0, P0 = 2 nk - 1 - i0 (what enters at the front on the way back)
            if (MODE == 3) {
                const int P0 = 2 * (int)nkp - 1 - (int)i0;
                const int lo = P0 - 15 > 0 ? P0 - 15 : 0;
                const u32 wq = (u32)(lo >> 4) < (u32)(NWL - 2) ? (u32)(lo >> 4) : (u32)(NWL - 2);
                const u64 win = ((u64)sw[(wq + 1) * 64 + lane] << 32) | sw[wq * 64 + lane];
                const int sft = P0 - 15 - 16 * (int)wq;  // -15 .. (beyond 15: no position of this block lies inside the read)
                cb2 = sft >= 32 ? 0u : sft >= 0 ? (u32)(win >> (2 * sft)) : (u32)(win << (-2 * sft));
            }
            const u32 cinb = __builtin_amdgcn_alignbit(sw[((t0 >> 4) + 1) * 64 + lane], sw[(t0 >> 4) * 64 + lane], (t0 & 15) * 2);
            const u32 olo = sw[(p0 >> 4) * 64 + lane], ohi = sw[((p0 >> 4) + 1) * 64 + lane];
            const u32 coutb = i0 ? __builtin_amdgcn_alignbit(ohi, olo, (p0 & 15) * 2) : (olo << 2);
            u32x4 xs[16];
            if (MODE < 2) {
#pragma unroll
                for (int o = 0; o < 16; ++o) {
                    const u32 ia = (o >= 2 ? (cinb >> (2 * o - 4)) : (cinb << (4 - 2 * o))) & 0x30u;
                    u32 ib = (o >= 3 ? (coutb >> (2 * o - 6)) : (coutb << (6 - 2 * o))) & 0xC0u;
                    if (o == 0) ib = i0 ? ib : 0x100u;  // very first k-mer: nothing leaves
                    xs[o] = *reinterpret_cast<LDSQ const u32x4 *>(lq + (ia | ib));
                }
            }
#pragma unroll
            for (int o = 0; o < 16; ++o) {
                u32 hl, hh;
                if (MODE == 3) {
                    const u32 j = i0 + (u32)o;  // value j of the run
                    const bool fwd = j < nkp, back = j > nkp;  // (j == nkp: the reverse-complement code of the last position, as the forward walk left it)
                    const u32 b = (cinb >> (2 * o)) & 3, bb = (cb2 >> (2 * (15 - o))) & 3;
                    const u64 ncode = ((code << 2) | b) & cmask;                  // iterator.go:736
                    const u64 nrc_f = (rc >> 2) | ((u64)cm(b) << sh2);           // (iterator.go:740's roll, on the paired letter)
                    const u64 nrc_b = ((rc << 2) & cmask) | (u64)cm(bb);          // the same code one position to the LEFT
                    code = fwd ? ncode : code;
                    rc = fwd ? nrc_f : back ? nrc_b : rc;
                    const u64 out = fwd ? code : rc;
                    hl = (u32)out;
                    hh = (u32)(out >> 32);
                } else if (MODE == 4) {
                    const u32 b = (cinb >> (2 * o)) & 3;
                    code = ((code << 2) | b) & cmask;
                    hl = (u32)code;
                    hh = (u32)(code >> 32);
                } else if (MODE == 2) {
                    const u32 b = (cinb >> (2 * o)) & 3;
                    code = ((code << 2) | b) & cmask;                   // iterator.go:736
                    rc = (rc >> 2) | ((u64)(b ^ 3u) << sh2);            // iterator.go:740
                    const lmask rev = lt64((u32)rc, (u32)(rc >> 32), (u32)code, (u32)(code >> 32));  // iterator.go:754
                    hl = sel(rev, (u32)rc, (u32)code);
                    hh = sel(rev, (u32)(rc >> 32), (u32)(code >> 32));
                } else {
                    roll(xs[o]);
                    hl = fl;
                    hh = fh_;
                    if (MODE == 1) {
                        const lmask rev = lt64(rl, rh_, fl, fh_);
                        hl = sel(rev, rl, fl);
                        hh = sel(rev, rh_, fh_);
                    }
                }
                if constexpr (CP) *reinterpret_cast<LDSQ u64 *>(lq + (PAR ? wa1[o] : wa0[o])) = ((u64)hh << 32) | hl;
                else *reinterpret_cast<LDSQ u64 *>(myrow0 + o * 8) = ((u64)hh << 32) | hl;
            }
            wave_sync_lds();
            // flush: 8 x (16 bytes per lane = 128 bytes per read); runs are padded to whole lines, so a lane either
            // stores its full 16 bytes or nothing (the padding receives whatever the tile holds)
            if constexpr (CP) {
#ifndef NTCP_NOBOUNDARY
                if (i0 == 0) {
#pragma unroll
                    for (int o = 0; o < 16; ++o) hv[o] = *reinterpret_cast<LDSQ const u64 *>(lq + wa0[o]);
                }
#endif
                u32x4 tv[8];
#pragma unroll
                for (int rr = 0; rr < 8; ++rr)
                    tv[rr] = *reinterpret_cast<LDSQ const u32x4 *>(lq + 512 + (rr * 8 + (lane >> 3)) * (TL * 8) + PAR * 128 + (lane & 7) * 16);
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {  // the line leaves if it lies inside the read (its first and its last line wait for the neighbours)
#ifdef NTCP_ALLSTORE  // dev (timing only)
                    if (rnk[rr]) {
#else
                    if (rnk[rr] && i0 >= re[rr] && i0 - re[rr] + 16u <= rnk[rr]) {
#endif
                        u64x2_a8 vv;
                        vv.a = ((u64)tv[rr].y << 32) | tv[rr].x;
                        vv.b = ((u64)tv[rr].w << 32) | tv[rr].z;
                        nt_store_u64x2(a.hash + roff[rr] + i0, vv.a, vv.b);
                    }
                }
            } else {
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
            }
            wave_sync_lds();
        };
        if constexpr (CP) {
            for (u32 i0 = 0; i0 < nk_max; i0 += 32) {
                blockfn(std::integral_constant<int, 0>{}, i0);
                if (i0 + 16 < nk_max) blockfn(std::integral_constant<int, 1>{}, i0 + 16);
            }
        } else {
            for (u32 i0 = 0; i0 < nk_max; i0 += 16) blockfn(std::integral_constant<int, 0>{}, i0);
        }
#ifndef NTCP_NOBOUNDARY  // dev (timing only)
        if constexpr (CP) {
            // the lines shared by two reads: tail of row (from its ring) + head of row + 1 (from the heads, which take the words' place)
            LDSQ char *const hd = lq + SW_OFF;
#pragma unroll
            for (int o = 0; o < 16; ++o) *reinterpret_cast<LDSQ u64 *>(hd + (lane * 17 + o) * 8) = hv[o];
            wave_sync_lds();
            const u32 NK = nk_max;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const u32 row = (u32)(rr * 8 + (lane >> 3));
                const u32 end = (row + 1u) * NK;              // unit-relative index one past the row's last value
                const u32 ls = (end - 1u) & ~15u;             // the line that holds that value
                const u32 g = ls + (u32)(lane & 7) * 2;       // this lane's two values of it
                u64 v[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const u32 gi = g + (u32)q;
                    const bool tail = gi < end;
                    const u32 tcol = (gi - row * NK + re[rr]) & 31u;              // (ring column = value index + the row's shift; the line begins inside the row: NK >= 32)
                    const u32 hcol = tail ? 0u : (gi - end);                      // < 16
                    const u32 hrow = row + 1u < 64u ? row + 1u : 63u;
                    const u64 tv_ = *reinterpret_cast<LDSQ const u64 *>(lq + 512 + row * (TL * 8) + tcol * 8u);
                    const u64 hv_ = *reinterpret_cast<LDSQ const u64 *>(hd + (hrow * 17u + hcol) * 8u);
                    v[q] = tail ? tv_ : hv_;
                }
                if (rnk[rr] && (end & 15u)) nt_store_u64x2(a.hash + base + g, v[0], v[1]);
            }
            wave_sync_lds();
        }
#endif
      }
    }
}

// ---- dispatch table --------------------------------------------------------------------
#ifdef BSK_IMPL_MINIMIZER  // dispatch functions: compiled in the family's own translation unit
#ifndef BSK_FAST_WS  // (dev builds narrow the list: -D'BSK_FAST_WS(X)=X(11)')
#define BSK_FAST_WS(X) \
    X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) \
    X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32)
#endif

bool fast_minimizer_supported(int w) {
    switch (w) {
#define X(WW) case WW:
        BSK_FAST_WS(X)
#undef X
        return true;
        default: return false;
    }
}

template <bool POS16>
static inline int fast_minimizer_blocks_per_cu_t(int w) {
    int nb = 0;
    hipError_t e = hipErrorInvalidValue;
    switch (w) {
#define X(WW) \
    case WW: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_minimizer_fast<WW, BSK_FAST_CAP, POS16>, 64, 0); break;
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
int fast_minimizer_blocks_per_cu(int w) { return fast_minimizer_blocks_per_cu_t<true>(w); }

// staged positions are 15 bit + strand: the fast path takes reads shorter than 32768 bases (longer: generic kernel)
void fast_minimizer_launch(int w, int grid, hipStream_t stream, const KArgs &a) {
    switch (w) {
#define X(WW) \
    case WW: hipLaunchKernelGGL((k_minimizer_fast<WW, BSK_FAST_CAP, true>), dim3(grid), dim3(64), 0, stream, a); break;
        BSK_FAST_WS(X)
#undef X
        default: break;
    }
}

#endif  // BSK_IMPL_MINIMIZER

#ifdef BSK_IMPL_DENSE  // k_minimizer_dense<W>: its own translation unit (k_minimizer_dense.hip)
#define BSK_DENSE_WS(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)
bool dense_minimizer_supported(int w) { return w >= 1 && w <= 16; }  // (w = 1: every k-mer with its position -- sketch.go:218-222 -- ran on the general kernel until round 5)
int dense_minimizer_blocks_per_cu(int w) {
    int nb = 0;
    hipError_t e = hipErrorInvalidValue;
    switch (w) {
#define X(WW) \
    case WW: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_minimizer_dense<WW>, 64, 0); break;
        BSK_DENSE_WS(X)
#undef X
        default: break;
    }
    if (e != hipSuccess || nb < 1) {
        (void)hipGetLastError();
        nb = 1;
    }
    return nb;
}
void dense_minimizer_ascii_launch(int w, int grid, hipStream_t stream, const KArgs &a) {  // the side launch over a.subset, from ASCII
    switch (w) {
#define X(WW) \
    case WW: hipLaunchKernelGGL((k_minimizer_dense<WW, false, true>), dim3(grid), dim3(64), 0, stream, a); break;
        BSK_DENSE_WS(X)
#undef X
        default: break;
    }
}
int dense_minimizer_ascii_blocks_per_cu(int w) {
    int nb = 0;
    hipError_t e = hipErrorInvalidValue;
    switch (w) {
#define X(WW) \
    case WW: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_minimizer_dense<WW, false, true>, 64, 0); break;
        BSK_DENSE_WS(X)
#undef X
        default: break;
    }
    if (e != hipSuccess || nb < 1) {
        (void)hipGetLastError();
        nb = 1;
    }
    return nb;
}
void dense_minimizer_launch(int w, int grid, hipStream_t stream, const KArgs &a) {
    switch (w) {
#define X(WW) \
    case WW: hipLaunchKernelGGL((k_minimizer_dense<WW>), dim3(grid), dim3(64), 0, stream, a); break;
        BSK_DENSE_WS(X)
#undef X
        default: break;
    }
}
#endif  // BSK_IMPL_DENSE

}  // namespace bsk
