// kernels_syncmer_pk.hpp -- k_syncmer_pk<W = k - s>: the reference's window-bounded closed syncmer (sketch.go:312-477; closed form and
// the fused one-pass structure: kernels_syncmer.hpp) on the packed window machine of kernels_pk.hpp.
//
//   * an s-mer is ONE 32-bit word: the upper 27 bits of its canonical hash | its offset O inside its block of W s-mers.  Prefix
//     minimum, suffix minimum, their combination M[j] (leftmost minimum of s-mers [j, j+W-1]) and the choice between the two halves
//     of the 2W window (M[idx] from W steps ago, kept in D[O], against M[idx+W]) are one v_min_u32 each;
//   * the selected k-mer position b is the winner's s-mer position (left half) or that minus W (right half): the same residue
//     modulo W either way, and block offsets ARE residues modulo W (blocks start at multiples of W).  The not yet emitted selections
//     lie in [idx, idx+W-1] -- W consecutive positions, distinct residues -- so they are W bits: sel |= 1 << O_winner (one
//     v_lshl_or_b32), and "is position idx selected" is bit (idx mod W), a compile-time bit number at every step of the unrolled
//     block; the bit is cleared when idx passes.  (The reference's emit-when-reached queue, sketch.go:424-475, as a bitmask.)
//   * S[] and D[] are one register per slot instead of three: the kernel needs ~150 VGPRs instead of 242 -- THREE waves per SIMD --
//     and a read selects 7 positions (k=31 s=11, 150 bp), so 24 rows per pair of lanes stage a unit in 9 KB of LDS: twelve waves per
//     CU fit, where k_syncmer_fast ran eight;
//   * exactness as in kernels_pk.hpp: the packed minimum is the 64-bit leftmost minimum unless two s-mers with equal 27-bit keys met
//     in a min operation (now four per step); such a unit, or one with a full staging column, goes to a list and k_syncmer_fast's
//     exact machine runs it afterwards (k_syncmer_fix), which also evaluates BSK_ST_FIRST_WINDOW_TIE.
// Reads of up to 16 (PKNW - 2) = 224 bases (pk_syncmer_max_bases); words, descriptors and waits as in k_minimizer_pk.
#pragma once
#include "kernels_pk.hpp"
#include "kernels_syncmer.hpp"

namespace bsk {

#ifndef BSK_SYNPK_ROWS
#define BSK_SYNPK_ROWS 23
#endif
#ifndef BSK_SYNPKL_ROWS
#define BSK_SYNPKL_ROWS 58
#endif
#ifndef BSK_SYNPKL_NW
#define BSK_SYNPKL_NW 32
#endif
// PR_: rows of a pair's staging column.  NW_: packed words of a read kept in registers.  LIM: the LDS a wavefront may take.
// DMA_: the next unit's words and descriptors travel global -> LDS (buffers WBUF / DBUF); otherwise global -> registers
template <int PR_, int NW_, int LIM, bool DMA_ = true>
struct SynPkLdsT {
    static constexpr int PR = PR_;
    static constexpr int NW = NW_;
    static constexpr bool DMA = DMA_;
    static constexpr int ROW = 33;
    static constexpr int TABK = 0;      // 20 x uint4 k-mer update table } hash phase
    static constexpr int TABS = 320;    // 20 x uint4 s-mer update table }
    static constexpr int NHEADS = (32 * (PR - 1)) / 64 + 2;
    static constexpr int HEADS = 0;     // } copy-out, laid over the two tables
    static constexpr int CTAB = 256;    // }
    static constexpr int SH = 768;
    static constexpr int SP = SH + (PR + 1) * ROW * 8;
    static constexpr int WBUF = (SP + (PR + 1) * ROW * 2 + 15) & ~15;  // u32x4 [NW / 4][64]: the NEXT unit's packed words (LDS-DMA)
    static constexpr int DBUF = WBUF + NW * 64 * 4;                    // u32 [2][64]: the next unit's descriptors, low and high words
    static constexpr int TOTAL = DMA ? DBUF + 512 : WBUF;              // short plan: 13 296 B, twelve waves per CU
    static_assert(NHEADS * 8 <= CTAB && CTAB + 512 <= SH && TABS + 320 <= SH && TOTAL <= LIM && NW % 4 == 0, "SynPkLds");
};
typedef SynPkLdsT<BSK_SYNPK_ROWS, PKNW, 13312> SynPkLds;              // k_syncmer_pk: reads of up to 224 bases, twelve waves per CU
// k_syncmer_pkl: reads of up to 480 bases and k - s up to 24 in 58-row columns, two waves per SIMD (round 4: until then reads that
// select more than ~10 positions or are longer than 224 bases, and k - s = 21..24, ran on the 64-bit machine k_syncmer_fast)
// (two waves per SIMD have the registers to take the next unit's words as k_minimizer_pk does, and the LDS the two buffers would take
// is 18 more rows at the same eight waves per CU -- seven ran 17 % slower, six 26 %: round 4, scripts/dev/perf_syn_long.py)
#ifndef BSK_SYNPKL_DMA
#define BSK_SYNPKL_DMA 0
#endif
typedef SynPkLdsT<BSK_SYNPKL_ROWS, BSK_SYNPKL_NW, 32768, BSK_SYNPKL_DMA != 0> SynPkLdsL;  // 58 rows: 20 240 B, eight waves per CU

// The next unit's packed words go global -> LDS directly (global_load_lds_dwordx4: lane i's 16 bytes land at base + 16 i), from inline
// asm: no VGPR destination that the compiler could spill or copy while the load is in flight (under this kernel's 168-register cap
// it did exactly that with k_minimizer_pk's register form), and no entry in its s_waitcnt bookkeeping (kernels_pk.hpp: PkMin::word2).
// The caller waits with vmcnt(0) before the unit's copy-out; M0 (compiler-reserved) is saved and restored inside the statement.
template <int NQ>
__device__ __forceinline__ void synpk_dma_words(const u32 *gsrc, u32 lds_dst) {
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        u32 keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(gsrc + 4 * j), "s"(lds_dst + 1024u * (u32)j)
                     : "memory");
    }
}

__device__ __forceinline__ void synpk_dma_desc(const u64 *gsrc, u32 lds_dst) {  // the lane's descriptor: two dwords, lane-linear each
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        u32 keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(reinterpret_cast<const u32 *>(gsrc) + j), "s"(lds_dst + 256u * (u32)j)
                     : "memory");
    }
}

// register form of the prefetch (plans without the LDS buffers): NQ x 4 words by loads the s_waitcnt pass does not see (kernels_pk.hpp)
template <int NQ>
struct SynWords {
    u32x4 q[NQ];
};
template <int NQ>
__device__ __forceinline__ SynWords<NQ> syn_load_words(const u32 *p) {
    SynWords<NQ> r;
    static_assert(NQ == 4 || NQ == 6 || NQ == 8, "SynWords");
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
                 "global_load_dwordx4 %2, %4, off offset:32\n\tglobal_load_dwordx4 %3, %4, off offset:48"
                 : "=&v"(r.q[0]), "=&v"(r.q[1]), "=&v"(r.q[2]), "=&v"(r.q[3])
                 : "v"(p));
    if constexpr (NQ >= 6) asm volatile("global_load_dwordx4 %0, %2, off offset:64\n\tglobal_load_dwordx4 %1, %2, off offset:80" : "=&v"(r.q[4]), "=&v"(r.q[5]) : "v"(p));
    if constexpr (NQ >= 8) asm volatile("global_load_dwordx4 %0, %2, off offset:96\n\tglobal_load_dwordx4 %1, %2, off offset:112" : "=&v"(r.q[6]), "=&v"(r.q[7]) : "v"(p));
    return r;
}
template <int NQ>
__device__ __forceinline__ void syn_wait_loads(SynWords<NQ> &p, u64 &d0, u32 &f) {
    if constexpr (NQ == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(p.q[0]), "+v"(p.q[1]), "+v"(p.q[2]), "+v"(p.q[3]), "+v"(d0), "+v"(f)::"memory");
    else if constexpr (NQ == 6)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(p.q[0]), "+v"(p.q[1]), "+v"(p.q[2]), "+v"(p.q[3]), "+v"(p.q[4]), "+v"(p.q[5]), "+v"(d0), "+v"(f)::"memory");
    else
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(p.q[0]), "+v"(p.q[1]), "+v"(p.q[2]), "+v"(p.q[3]), "+v"(p.q[4]), "+v"(p.q[5]), "+v"(p.q[6]), "+v"(p.q[7]), "+v"(d0), "+v"(f)::"memory");
}

struct SynPkTabs {  // lanes 0..19: the k-mer table's row, lanes 32..51: the s-mer table's
    u32x4 row;
    u32 at;
    __device__ __forceinline__ void init(int k, int s, int lane) {
        row = (u32x4){0, 0, 0, 0};
        at = 0xffffffffu;
        const int t = lane & 31;
        if (t < 20) {
            const int kk = lane < 32 ? k : s;
            const unsigned out = (unsigned)t >> 2, in = (unsigned)t & 3u;
            u64 f = seed_fwd_code(in);
            u64 r = rol64(seed_rev_code(in), (unsigned)(kk - 1));
            if (out < 4) {
                f ^= rol64(seed_fwd_code(out), (unsigned)kk);
                r ^= ror64(seed_rev_code(out), 1);
            }
            row = (u32x4){(u32)f, (u32)(f >> 32), (u32)r, (u32)(r >> 32)};
            at = (u32)((lane < 32 ? SynPkLds::TABK : SynPkLds::TABS) + t * 16);
        }
    }
    __device__ __forceinline__ void write(LDSQ char *ldsq) const {
        if (at != 0xffffffffu) *reinterpret_cast<LDSQ u32x4 *>(ldsq + at) = row;
        wave_sync_lds();
    }
};

// table row offsets of up to 32 steps: nibble j of E / O (two words each) = (out << 2 | in) of step 2j / 2j+1
struct Nib32 {
    u32 e0, o0, e1, o1;
    __device__ __forceinline__ void set(const Codes32 &in, const Codes32 &out) {
        e0 = (in.lo & 0x33333333u) | ((out.lo & 0x33333333u) << 2);
        o0 = ((in.lo >> 2) & 0x33333333u) | (out.lo & 0xCCCCCCCCu);
        e1 = (in.hi & 0x33333333u) | ((out.hi & 0x33333333u) << 2);
        o1 = ((in.hi >> 2) & 0x33333333u) | (out.hi & 0xCCCCCCCCu);
    }
    template <int O>
    __device__ __forceinline__ u32 off() const {
        constexpr int q = O & 15, j = q >> 1;
        const u32 src = O < 16 ? ((q & 1) ? o0 : e0) : ((q & 1) ? o1 : e1);
        return (j >= 1 ? (src >> (4 * j - 4)) : (src << 4)) & 0xF0u;
    }
};

typedef u32 u32x32 __attribute__((ext_vector_type(32)));
template <int N>
struct SynVec {
    static_assert(N == 16 || N == 24 || N == 32, "SynVec");
    typedef typename std::conditional<N == 16, u32x16, u32x32>::type type;  // (no 24-register class: the tuple is 32 wide)
};
// SEL (k_syncmer_sel, kernels_syncmer_sel.hpp): the SELECTION alone -- no k-mer hash, no staging: bit O of a block's word says whether
// idx = idx0 + O is selected, the word leaves to HBM at the end of the block and a second pass hashes the selected k-mers only.
// SEL = 2 (k_syncmer_pf, kernels_syncmer_pf.hpp): the same, but the word stays in LDS (row i0 / W - 1 of LY::MASK, laid over the parked
// suffix minima, which block 1 has read by then) and the unit's own emit phase hashes what was selected -- one pass over HBM.
template <int W, class LY_ = SynPkLds, int SEL = 0>
struct SynPk {
    typedef LY_ LY;
    static constexpr int NW = LY::NW;
    LDSQ char *lds;
    int k, s, lane;
    u32 bsel, nsel;   // SEL: the current block's selection bits, the lane's count so far
    u32 *gmask;       // SEL: this lane's column of the unit's mask rows (row m is 64 words further)
    u32 end_plus1;  // number of windows of this lane (end + 1), 0: the lane does not stage
    typename SynVec<NW>::type wr;  // the read's first NW packed words
    u32 kfl, kfh, krl, krh, sfl, sfh, srl, srh;
    u32 S[W], D[W], P;
    u32 selm, tmin, slot, spare, park0;
    int sstep;

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
    __device__ __forceinline__ void tie(u32 a, u32 b) {
#ifndef SYNPK_NOTIE  // (dev knock-out, timing only)
        const u32 d = a ^ b;
        tmin = tmin < d ? tmin : d;
#else
        (void)a, (void)b;
#endif
    }
    // 32 codes from base position p0 (wave-uniform; positions beyond the 16 words read as the last words: never a valid step's)
    __device__ __forceinline__ Codes32 codes(u32 p0) const {
        u32 wi = (u32)__builtin_amdgcn_readfirstlane((int)(p0 >> 4));
        wi = wi < (u32)(NW - 3) ? wi : (u32)(NW - 3);
        Codes32 c;
        const u32 w0 = wr[wi], w1 = wr[wi + 1], w2 = wr[wi + 2], sh = (p0 & 15) * 2;
        c.lo = __builtin_amdgcn_alignbit(w1, w0, sh);
        c.hi = __builtin_amdgcn_alignbit(w2, w1, sh);
        return c;
    }

    // MODE 0: s-mer block 0 (priming).  MODE 1: block 1 (the first 2W window completes at its last offset).
    // MODE 2: steady state, every lane's windows exist.  MODE 3: steady state, windows end per lane.
    template <int MODE>
    __device__ __forceinline__ void block(u32 i0, bool suffix) {
        constexpr bool FUSED = MODE >= 2;
        Codes32 sin = codes(i0 + (u32)s - 1), sout = codes(i0 ? i0 - 1 : 0);
        if (MODE == 0) {  // block 0: offset o >= 1 sees base o-1 (offset 0 takes the "nothing leaves" row)
            const u64 v = (((u64)sout.hi << 32) | sout.lo) << 2;
            sout.lo = (u32)v;
            sout.hi = (u32)(v >> 32);
        }
        Nib32 ns, nk;
        ns.set(sin, sout);
        const u32 idx0 = i0 - (u32)(2 * W - 1);  // idx of offset 0 (MODE >= 2)
        if (SEL != 0) {
        } else if (FUSED) {
            nk.set(codes(idx0 + (u32)k - 1), codes(idx0 - 1));
        } else if (MODE == 1) {  // only offset W-1 is a fused step: idx = 0, incoming base k-1, nothing leaves
            Codes32 kin = codes((u32)k - 1), z;
            const u64 v = (((u64)kin.hi << 32) | kin.lo) << (2 * (W - 1));
            kin.lo = (u32)v;
            kin.hi = (u32)(v >> 32);
            z.lo = z.hi = 0;
            nk.set(kin, z);
        }
        u32 vb = 0;
        // MODE 3: s-mers beyond the lane's own end (the lane runs on to the wave's longest read) take made-up keys that differ from
        // each other wherever two of them can meet in a min operation -- offset and block parity in the key's low bits, ones above:
        // they lose against every real s-mer and never tie.  What lies behind a read are the padding bases of its last word (code 0 = A)
        // and the next read: twelve or more padding bases are two or more IDENTICAL poly-A s-mers, a key tie by construction -- the read
        // went to the list for nothing.  Length-binned units keep the overrun below a class width; tiles and batches in batch order do
        // not: 450-base reads cut into tiles of 333 + 226 bases (14 padding bases) listed every second tile, the list filled up and
        // the call fell back to k_syncmer_fast after a wasted run (round 4, scripts/dev/perf_midlen.py).
        int nvs = W;          // s-mers of this block the lane has
        u32 badkey = 0;
        if (MODE == 3) {  // bit o: window idx0 + o exists for this lane
            const int left = (int)end_plus1 - (int)idx0;
            const u32 nv = (u32)(left < 0 ? 0 : left > W ? W : left);
            vb = nv >= 32u ? 0xffffffffu : (1u << nv) - 1u;
            nvs = end_plus1 ? (int)end_plus1 + (2 * W - 1) - (int)i0 : W;  // (ns = end + 1 + 2W - 1; lanes without a read are never listed)
            badkey = 0xfffff000u | (((i0 / (u32)W) & 1u) << 10);
        }
        // table rows: XC steps' worth (both tables) are requested one chunk ahead of their use -- a row requested where it is rolled in
        // exposes the LDS latency twice per step.  One step ahead is enough at twelve waves per CU, and the registers matter more:
        // chunks of 4 / 3 / 2 / 1 steps spill 172 / 128 / 72 / 12 bytes under the 168-VGPR cap and run 846 / 934 / 939 / 968 Gbases/s
        // (spills inside the block loop are HBM traffic: 43 GB per launch against 17 GB algorithmic with chunks of 4)
#ifndef SYNPK_XC
#define SYNPK_XC 1
#endif
#ifndef SYNPKL_XC
#define SYNPKL_XC 1
#endif
        constexpr int XC = LY::NW > PKNW ? SYNPKL_XC : SYNPK_XC;  // (the long plan has the registers for more)
        u32x4 xs[W], xk[W];
        auto fetch = [&](auto o0c) {
            constexpr int O0 = decltype(o0c)::value;
            pk_unroll<XC>([&](auto jc) {
                constexpr int O = O0 + decltype(jc)::value;
                if constexpr (O < W) {
                    u32 so = ns.template off<O>();
                    if (MODE == 0 && O == 0) so = 0x100u | (so & 0x30u);
                    xs[O] = tabs(so);
#ifndef SYNPK_NOK
                    if (SEL == 0 && (FUSED || (MODE == 1 && O == W - 1))) {
                        u32 ko = nk.template off<O>();
                        if (MODE == 1) ko = 0x100u | (ko & 0x30u);
                        xk[O] = tabk(ko);
                    }
#endif
                }
            });
        };
        fetch(std::integral_constant<int, 0>{});
        fetch(std::integral_constant<int, XC>{});
        pk_unroll<W>([&](auto oc) {
            constexpr int O = decltype(oc)::value;
            if constexpr (O > 0 && O % XC == 0 && O + XC < W) {
                __builtin_amdgcn_sched_barrier(0);  // keep the next chunk's reads here (hoisted, they are all live at once)
                fetch(std::integral_constant<int, O + XC>{});
            }
            // ---- s-mer i_s = i0 + O ----
            rolls(xs[O]);
            // the key is the HIGH word of the canonical hash min(fwd, rev): min(fwd.hi, rev.hi) exactly -- where the high words are equal the
            // minimum's high word is that value whichever strand wins -- one full-rate v_min_u32 instead of a 64-bit compare and a select
            const u32 sh_ = sfh < srh ? sfh : srh;
            u32 pk;
            asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(pk) : "v"(sh_), "s"(0xffffffe0u), "n"(O));
            if (MODE == 3) pk = nvs > O ? pk : (badkey | (u32)(O << 5) | (u32)O);
            if (O == 0) {
                P = pk;
            } else {
                tie(pk, P);
                P = P < pk ? P : pk;
            }
            if (MODE >= 1 || O == W - 1) {
                u32 M = P;  // leftmost min of s-mers [i_s-W+1, i_s]
                if constexpr (MODE >= 1 && O != W - 1) {
                    tie(P, S[O + 1]);
                    M = P < S[O + 1] ? P : S[O + 1];
                }
                if (FUSED || (MODE == 1 && O == W - 1)) {
                    // ---- fused step: idx = i_s - 2W + 1, residue (O + 1) mod W ----
                    constexpr int RS = (O + 1) % W;
                    tie(M, D[O]);
                    const u32 win = M < D[O] ? M : D[O];  // (a tie: the exact machine decides)
                    selm |= 1u << (win & 31u);            // v_lshl_or_b32: the winner's residue
                    if constexpr (SEL != 0) {
                        u32 b = (selm >> RS) & 1u;
                        if (MODE == 3) b &= vb >> O;
                        selm &= ~(1u << RS);
                        bsel |= b << O;  // v_lshl_or_b32 (lanes without a read: masked when the word leaves)
                    } else {
#ifdef SYNPK_NOK  // dev knock-out (timing only): the s-mer machine alone -- no k-mer hash, no staging (what a selection-only first pass would cost)
                    u32 b = (selm >> RS) & 1u;
                    if (MODE == 3) b &= vb >> O;
                    if (MODE == 1) b = end_plus1 ? b : 0u;
                    selm &= ~(1u << RS);
                    asm("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(slot) : "v"(b), "v"(sstep));
#else
                    const u32 idx = FUSED ? idx0 + (u32)O : 0u;
                    rollk(xk[O]);
                    const lmask krev = lt64(krl, krh, kfl, kfh);
                    const u32 hl = sel(krev, krl, kfl), hh = sel(krev, krh, kfh);
                    const u32 ps = (sel01(krev) << 15) | idx;
                    u32 b = (selm >> RS) & 1u;
                    if (MODE == 3) b &= vb >> O;  // (bit 0 of the product; vb's other bits are masked by b)
                    if (MODE == 1) b = end_plus1 ? b : 0u;
                    selm &= ~(1u << RS);
                    const u32 addr = slot < spare ? slot : spare;
#ifndef SYNPK_NOSTAGE
                    *reinterpret_cast<LDSQ u64 *>(lds + LY::SH + addr) = ((u64)hh << 32) | hl;
                    *reinterpret_cast<LDSQ u16 *>(lds + LY::SP + (addr >> 2)) = (u16)ps;
#endif
                    asm("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(slot) : "v"(b), "v"(sstep));
#endif
                    }
                }
                D[O] = M;
            }
            S[O] = pk;
        });
        // BSK_ST_FIRST_WINDOW_TIE (kernels_fast.hpp, suffix_min_pass: "the minimum of s-mers [q, 2W) occurs twice" for some q) must
        // reach the exact machine for every read it would flag.  Ties inside a block meet in that block's suffix pass and ties
        // between the two blocks' minima in the first fused step; what no min operation sees is a suffix minimum S0[q] of block 0,
        // q > 0, equal to block 1's minimum T.  Block 0 parks its suffix minima in the lane's staging rows (empty but for the first
        // row until block 2) and block 1 compares them with T.
        constexpr int RB = LY::ROW * 8;
        const int park = sstep > 0 ? (int)slot + RB : sstep < 0 ? (int)slot - ((W + 1) / 2) * RB : (int)spare;  // rows 1.. upwards / below the top row
        if constexpr (SEL != 0) {  // (no staging rows: the lane's column of a [W][64] table)
            if (MODE == 1) {
#pragma unroll
                for (int q = 0; q < W; ++q) tie(*reinterpret_cast<LDSQ const u32 *>(lds + LY::PARK + q * 256 + lane * 4), P);
            }
            if (MODE >= 1) {  // the block's selection word leaves: row i0 / W - 1 of the unit's mask rows
                const u32 wsel = end_plus1 ? bsel : 0u;
                if constexpr (SEL == 2) *reinterpret_cast<LDSQ u32 *>(lds + LY::MASK + (i0 / (u32)W - 1u) * 256u + (u32)lane * 4u) = wsel;
                else gmask[(i0 / (u32)W - 1u) * 64u] = wsel;
                nsel += (u32)__builtin_popcount(wsel);
                bsel = 0;
            }
        } else
        if (MODE == 1) {
#pragma unroll
            for (int q = 0; q < W; ++q) tie(*reinterpret_cast<LDSQ const u32 *>(lds + LY::SH + park0 + (q >> 1) * RB + (q & 1) * 4), P);
        }
        if (suffix) {
#pragma unroll
            for (int q = W - 2; q >= 0; --q) {
                tie(S[q], S[q + 1]);
                S[q] = S[q] < S[q + 1] ? S[q] : S[q + 1];
            }
        }
        if (MODE == 0) {
            if constexpr (SEL != 0) {
#pragma unroll
                for (int q = 0; q < W; ++q) *reinterpret_cast<LDSQ u32 *>(lds + LY::PARK + q * 256 + lane * 4) = S[q];
            } else {
            park0 = (u32)park;
            if (sstep != 0) {
#pragma unroll
                for (int q = 0; q < W; ++q) *reinterpret_cast<LDSQ u32 *>(lds + LY::SH + park0 + (q >> 1) * RB + (q & 1) * 4) = S[q];
            }
            }
        }
    }

    // ns_max: wave maximum of the number of s-mers; nwin_min: wave minimum of end + 1 over the lanes that stage (uniform batches)
    __device__ __forceinline__ void run(u32 ns_max, u32 nwin_min, u32 slot0, int step, u32 col8) {
        kfl = kfh = krl = krh = sfl = sfh = srl = srh = 0;
        selm = 0;
        bsel = nsel = 0;
        tmin = 0xffffffffu;
        spare = (u32)(LY::PR * LY::ROW * 8) + col8;
        slot = slot0;
        sstep = step;
        for (int t0 = 0; t0 < s - 1; t0 += 16) {  // s-mer warm-up: four table rows in flight per trip
            const u32 word = wr[(u32)__builtin_amdgcn_readfirstlane(t0 >> 4)];
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
        for (int t0 = 0; SEL == 0 && t0 < k - 1; t0 += 16) {  // k-mer warm-up
            const u32 word = wr[(u32)__builtin_amdgcn_readfirstlane(t0 >> 4)];
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
        block<0>(0, true);
        block<1>(W, true);  // (a lane that is not short has at least 2W s-mers, sketch.go:149)
        for (u32 i0 = 2 * W; i0 < ns_max; i0 += W) {
            const bool more = i0 + W < ns_max;
            // every window of the block exists for every staging lane: idx0 + W - 1 < nwin_min
            if (i0 - (u32)(2 * W - 1) + (u32)W <= nwin_min) block<2>(i0, more);
            else block<3>(i0, more);
        }
    }
};

#define BSK_SYNPK_TIE 0x80000000u
#ifndef SYNPK_LB
#define SYNPK_LB 3
#endif
template <int W, class LY>
__device__ __forceinline__ void synpk_body(const KArgs &a, char *lds) {
    constexpr int NQ = LY::NW / 4;
    LDSQ char *const ldsq = (LDSQ char *)lds;
    const int lane = lane_id();
    SynPkTabs tabs;
    tabs.init(a.k, a.s, lane);
    const u64 slab = (u64)64 * BSK_SYN_CAP;
    const u32 col8 = (u32)(lane & 31) * 8u;
    constexpr u32 RB = (u32)(LY::ROW * 8);
    const u32 top = (u32)(LY::PR - 1) * RB + col8;  // the high lane's first slot
    u64 d_cur = 0, d_nx = 0;
    SynWords<NQ> pw_cur;  // (register form of the prefetch)
#pragma unroll
    for (int j = 0; j < NQ; ++j) pw_cur.q[j] = (u32x4){0, 0, 0, 0};
    bool have = false;
    const u32 lseg = a.fixcap / a.list_grid;  // this workgroup's segment of the list of reads for the exact machine (list_append)
    u32 lcur = 0;
    const u32 wbuf = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(size_t)(ldsq + LY::WBUF));  // LDS byte addresses of the two buffers
    const u32 dbuf = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(size_t)(ldsq + LY::DBUF));
    const u32 tku = a.tk ? a.tk : 8u;
    for (u32 unit = next_ticket(a.ticket, lane) * tku, uend = unit + tku; unit < a.nunits; ++unit, ({
             if (unit == uend) {
                 unit = next_ticket(a.ticket, lane) * tku;
                 uend = unit + tku;
             }
         })) {
        const u64 r = (u64)unit * 64 + lane;
        const bool nxt = unit + 1 != uend && unit + 1 < a.nunits;
        // The next unit's words and descriptors travel global -> LDS while this unit is hashed (indices beyond the batch are clamped to
        // its last read; what such a load brings is never used) and are waited for before the copy-out.
        const u64 rmax = a.n - 1;
        typename SynVec<LY::NW>::type wr;
        u64 d_n1;
        SynWords<NQ> pw_n1;  // (register form)
        u64 d_n2 = 0;
        u32 rfl;
        if constexpr (LY::DMA) {
        if (!have) {  // first unit of a ticket: nothing was requested ahead
            d_cur = a.desc[r < rmax ? r : rmax];
            synpk_dma_words<NQ>(a.words + (d_cur >> 24), wbuf);
            synpk_dma_desc(a.desc + (r + 64 < rmax ? r + 64 : rmax), dbuf);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        {
            const LDSQ u32x4 *wb = reinterpret_cast<const LDSQ u32x4 *>(ldsq + LY::WBUF) + lane;
            const LDSQ u32 *db = reinterpret_cast<const LDSQ u32 *>(ldsq + LY::DBUF) + lane;
            u32x4 wq[NQ];
#pragma unroll
            for (int j = 0; j < NQ; ++j) wq[j] = wb[64 * j];
            u32 dl = db[0], dh = db[64];
            // both buffers are read before the next loads overwrite them
            static_assert(NQ == 4 || NQ == 6 || NQ == 8, "synpk_body");
            if constexpr (NQ == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wq[0]), "+v"(wq[1]), "+v"(wq[2]), "+v"(wq[3]), "+v"(dl), "+v"(dh)::"memory");
            else if constexpr (NQ == 6) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wq[0]), "+v"(wq[1]), "+v"(wq[2]), "+v"(wq[3]), "+v"(wq[4]), "+v"(wq[5]), "+v"(dl), "+v"(dh)::"memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wq[0]), "+v"(wq[1]), "+v"(wq[2]), "+v"(wq[3]), "+v"(wq[4]), "+v"(wq[5]), "+v"(wq[6]), "+v"(wq[7]), "+v"(dl), "+v"(dh)::"memory");
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                wr[4 * j] = wq[j].x;
                wr[4 * j + 1] = wq[j].y;
                wr[4 * j + 2] = wq[j].z;
                wr[4 * j + 3] = wq[j].w;
            }
            d_n1 = ((u64)dh << 32) | dl;
        }
        synpk_dma_words<NQ>(a.words + (d_n1 >> 24), wbuf);
        synpk_dma_desc(a.desc + (r + 128 < rmax ? r + 128 : rmax), dbuf);
        // the read's input flags (batches packed from ASCII have them): a load after the copy-out would wait for its stores
        rfl = pk_load_u8(a.rflags ? a.rflags + (r < rmax ? r : rmax) : reinterpret_cast<const u8 *>(a.desc));
        } else {  // register form, as k_minimizer_pk: every load unconditional, words of unit N+1 and descriptor of unit N+2 requested here
            if (!have) {
                d_cur = pk_load_u64(a.desc + (r < rmax ? r : rmax));
                d_nx = pk_load_u64(a.desc + (r + 64 < rmax ? r + 64 : rmax));
                pk_wait_loads(d_cur, d_nx);
                pw_cur = syn_load_words<NQ>(a.words + (d_cur >> 24));
                u64 dz = d_cur;
                u32 fz = 0;
                syn_wait_loads<NQ>(pw_cur, dz, fz);
            }
            d_n1 = d_nx;
            pw_n1 = syn_load_words<NQ>(a.words + (d_n1 >> 24));
            d_n2 = pk_load_u64(a.desc + (r + 128 < rmax ? r + 128 : rmax));
            rfl = pk_load_u8(a.rflags ? a.rflags + (r < rmax ? r : rmax) : reinterpret_cast<const u8 *>(a.desc));
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                wr[4 * j] = pw_cur.q[j].x;
                wr[4 * j + 1] = pw_cur.q[j].y;
                wr[4 * j + 2] = pw_cur.q[j].z;
                wr[4 * j + 3] = pw_cur.q[j].w;
            }
        }
        const u64 d = d_cur;
        const u64 L = desc_len(a, d);
        const u64 ro = out_index(a, r, d);  // (length-binned batches: the read's own place in its chunk)
        const long long Lorig = (long long)L - a.circ_ext;
        const bool ok = r < a.n && Lorig >= 0 && Lorig >= 2LL * a.k - a.s - 1 && L >= (u64)a.k;  // sketch.go:149
        const u32 nwin = ok ? (u32)(L - 2 * (u64)a.k + a.s + 2) : 0u;                             // end + 1
        const u32 ns = ok ? (u32)(L - a.s + 1) : 0u;
        const u32 ns_max = wave_max_u32(ns);
        const u32 nwin_min = ~wave_max_u32(ok ? ~nwin : 0u);  // minimum over the lanes with a read
        u32 cnt = 0, tmin_lane = 0xffffffffu;
        if (ns_max) {
            tabs.write(ldsq);  // the previous copy-out's tables took their place
            SynPk<W, LY> sp;
            sp.lds = ldsq;
            sp.k = a.k;
            sp.s = a.s;
            sp.lane = lane;
            sp.end_plus1 = nwin;
            sp.wr = wr;
            const u32 spare = (u32)LY::PR * RB + col8;
            sp.run(ns_max, nwin_min, !ok ? spare : lane < 32 ? col8 : top, !ok ? 0 : lane < 32 ? (int)RB : -(int)RB, col8);
            if (ok) cnt = (lane < 32 ? sp.slot - col8 : top - sp.slot) / RB;
            tmin_lane = sp.tmin;
        }
        if constexpr (LY::DMA) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(rfl)::"memory");  // the next unit's words and the descriptors after them are in LDS
        } else {
            syn_wait_loads<NQ>(pw_n1, d_n2, rfl);  // the next unit's words and the descriptor after it, requested a whole hashing phase ago
            pw_cur = pw_n1;
            d_nx = d_n2;
        }
        d_cur = d_n1;
        have = nxt;
        // reads the exact machine must run: two equal 27-bit keys met in one of their min operations (a few per cent of the reads at
        // s = 11: an s-mer and its reverse complement have the SAME canonical hash, and 11-mers that overlap their own reverse
        // complement are common), or their staging column filled up.  They go to a list of READS (k_syncmer_fast<W, true> gathers 64
        // of them per wavefront); the unit's other lanes leave normally.
        const u32 cnt_pair = cnt + (u32)__builtin_amdgcn_ds_bpermute((lane ^ 32) * 4, (int)cnt);
        u64 redo = __builtin_amdgcn_ballot_w64((ok && tmin_lane < 32u) || cnt_pair >= (u32)LY::PR);
#if defined(SYNPK_NOFB)
        redo = 0;
#endif
        if (redo) {
            list_append(a, reinterpret_cast<u32 *>(a.fixlist), lseg, lcur, redo, lane, r);
            if ((redo >> lane) & 1) cnt = 0;
        }
        const u32 incl = wave_incl_scan_u32(cnt, lane);
        const u32 excl = incl - cnt;
        const u32 T = wave_bcast_u32(incl, 63);
        const u64 base = (u64)unit * slab;
#ifndef SYNPK_NOCOPY
        if (T) pk_copyout<4, LY>(lds, lane, cnt, excl, T, base, a);
#endif
        if (r < a.n && !((redo >> lane) & 1)) {
            a.refs[ro] = ((base + excl) << 24) | cnt;
            u8 sbyte = ok ? BSK_ST_OK : BSK_ST_SHORT;
            if (ok && a.rflags) sbyte |= (u8)rfl;
            a.status[ro] = sbyte;
        }
    }
    list_close(reinterpret_cast<u32 *>(a.fixlist), lseg, lcur, lane);
}

template <int W>
__global__ __launch_bounds__(64, SYNPK_LB) void k_syncmer_pk(KArgs a) {  // three waves per SIMD: at most 168 VGPRs
    __shared__ __attribute__((aligned(16))) char lds[SynPkLds::TOTAL];
    synpk_body<W, SynPkLds>(a, lds);
}
// the same machine with room for longer reads: BSK_SYNPKL_NW words in registers (480 bases), BSK_SYNPKL_ROWS rows per pair of reads,
// two waves per SIMD (up to 256 VGPRs), eight waves per CU; also k - s = 21..24 (their first-window test parks 2 x 12 rows)
template <int W>
__global__ __launch_bounds__(64, 2) void k_syncmer_pkl(KArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[SynPkLdsL::TOTAL];
    synpk_body<W, SynPkLdsL>(a, lds);
}

#ifndef BSK_SYNPK_WS
#define BSK_SYNPK_WS(X) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20)
#endif
#ifndef BSK_SYNPKL_WS
#define BSK_SYNPKL_WS(X) BSK_SYNPK_WS(X) X(21) X(22) X(23) X(24)
#endif
// three translation units that compile side by side (clean build: 30 s + 30 s + 70 s instead of 113 s in one): k_syncmer_pkl.hip has the long
// plan's kernels, k_syncmer_fix.hip the fix pass k_syncmer_fast<W, true> of both plans, k_syncmer_pk.hip everything else
int pkl_syncmer_blocks_per_cu(int w);
void pkl_syncmer_launch(int w, int grid, hipStream_t stream, const KArgs &a);
void pk_syncmer_fix_launch(int w, int fix_grid, hipStream_t stream, const KArgs &a);
#ifdef BSK_IMPL_SYNPKL
int pkl_syncmer_blocks_per_cu(int w) {
    int nb = 0;
    hipError_t e = hipErrorInvalidValue;
#define X(WW) \
    if (w == WW) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_syncmer_pkl<WW>, 64, 0);
    BSK_SYNPKL_WS(X)
#undef X
    if (e != hipSuccess || nb < 1) {
        (void)hipGetLastError();
        nb = 1;
    }
    return nb;
}
void pkl_syncmer_launch(int w, int grid, hipStream_t stream, const KArgs &a) {
#define X(WW) \
    if (w == WW) hipLaunchKernelGGL((k_syncmer_pkl<WW>), dim3(grid), dim3(64), 0, stream, a);
    BSK_SYNPKL_WS(X)
#undef X
}
#endif  // BSK_IMPL_SYNPKL
#ifdef BSK_IMPL_SYNFIX
void pk_syncmer_fix_launch(int w, int fix_grid, hipStream_t stream, const KArgs &a) {
#define X(WW) \
    if (w == WW) hipLaunchKernelGGL((k_syncmer_fast<WW, true>), dim3(fix_grid), dim3(64), 0, stream, a);
    BSK_SYNPKL_WS(X)
#undef X
}
#endif  // BSK_IMPL_SYNFIX
#ifdef BSK_IMPL_SYNPK
// short plan: w <= 20 (the first-window tie test parks W suffix minima in the lane's half of a 23-row staging column); long plan: w <= 24
bool pk_syncmer_supported(int w, bool lng) {
#define X(WW) \
    if (w == WW) return true;
    if (lng) {
        BSK_SYNPKL_WS(X)
    } else {
        BSK_SYNPK_WS(X)
    }
#undef X
    return false;
}
static_assert(SynPkLdsL::NW <= 32 && SynPkLds::NW <= 32, "biosketch.hip: kMaxPrefetchWords (pad_words) must cover the widest register prefetch");
u32 pk_syncmer_max_bases(bool lng) { return 16u * (u32)((lng ? SynPkLdsL::NW : SynPkLds::NW) - 2); }  // the words a lane keeps in registers
u32 pk_syncmer_pair_rows(bool lng) { return (u32)(lng ? SynPkLdsL::PR : SynPkLds::PR); }
int pk_syncmer_blocks_per_cu(int w, bool lng) {
    if (lng) return pkl_syncmer_blocks_per_cu(w);
    int nb = 0;
    hipError_t e = hipErrorInvalidValue;
#define X(WW) \
    if (w == WW) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_syncmer_pk<WW>, 64, 0);
    BSK_SYNPK_WS(X)
#undef X
    if (e != hipSuccess || nb < 1) {
        (void)hipGetLastError();
        nb = 1;
    }
    return nb;
}
void pk_syncmer_launch(int w, bool lng, int grid, int fix_grid, hipStream_t stream, const KArgs &a) {
    if (lng) pkl_syncmer_launch(w, grid, stream, a);
#define X(WW) \
    if (w == WW && !lng) hipLaunchKernelGGL((k_syncmer_pk<WW>), dim3(grid), dim3(64), 0, stream, a);
    BSK_SYNPK_WS(X)
#undef X
    pk_syncmer_fix_launch(w, fix_grid, stream, a);
}
#endif  // BSK_IMPL_SYNPK

}  // namespace bsk
