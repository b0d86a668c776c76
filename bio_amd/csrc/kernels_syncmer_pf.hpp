// kernels_syncmer_pf.hpp -- k_syncmer_pf<W = k - s>: the packed syncmer machine with the emit FUSED into the unit (round 6).
//
// NextSyncmer (sketches/sketch.go:312-477) emits the canonical hash of a k-mer for ~1.5 / (k - s + 1) of the windows -- 7.1 of the 101
// windows of a 150-base read at k = 31, s = 11.  k_syncmer_pk rolls the k-mer hash over EVERY position, canonicalises and stages it,
// because a selection is known 2(k - s) - 1 steps late: a third of its instructions and all of its staging (timing build without:
// 1 584 against 974 Gbases/s).  Round 5 split the work into two kernels (kernels_syncmer_sel.hpp): the selection alone ran at
// 1 590 Gbases/s, but the second pass re-loaded descriptors, masks and words from HBM behind its own stores and lost (869).  Here the
// second pass is the END OF EVERY UNIT of the first, while everything it needs is still on the chip:
//
//   hash phase   SynPk<W, LY, SEL = 2>: the s-mer window machine alone.  One selection word per block of W windows and lane goes to LDS
//                (row i0 / W - 1 of MASK; the rows lie over the parked suffix minima of the first-window test, read by then).
//   expand       every lane's words -> the unit's tuple list FLAT[excl + j] = (lane << 8 | idx): a loop over the set bits of the rows
//   emit         the lanes' packed words go back from registers to LDS (EBUF, over the dead mask rows), then ONE LANE PER TUPLE, 64 tuples
//                per round: the k bases from idx on, ntHash FROM SCRATCH -- three bases per table row (64 rows of (fwd, rev) contributions;
//                ntHash is XOR-linear in its bases: fwd by Horner's rule, rev with the accumulator rotated the other way so that every
//                step rotates by a constant) -- canonical select, and the tuples leave as whole rows of 64 consecutive outputs straight
//                from registers: no staging columns, no copy-out, no column that fills up (a unit's slab is k_syncmer_pk's).
//
// Loads and the one vmcnt as in k_syncmer_pk: the next unit's words and descriptors travel global -> LDS (WBUF / DBUF) while this unit is
// hashed and are waited for before this unit's stores.  Reads with a 27-bit key tie go to the list of the exact machine
// (k_syncmer_fast<W, true>) as before; so do the last reads of a unit that selects more than BSK_PF_TCAP positions.
// LDS: 13 184 B per wavefront -- twelve waves per CU, three per SIMD (<= 168 VGPRs); k_syncmer_pfl (reads of up to 480 bases): 14 208 B, two per SIMD.
#pragma once
#include "kernels_syncmer_pk.hpp"

namespace bsk {

typedef u32 u32x2 __attribute__((ext_vector_type(2)));

// NW_: packed words of a read a lane keeps (16: reads of up to 224 bases, the next unit's words travel global -> LDS; 32: up to 480 bases,
// the next unit's words travel global -> registers as in k_syncmer_pkl).  MROWS_: rows of 256 bytes of the region the parked suffix
// minima (W rows), the selection words (one row per block) and, in the emit phase, the unit's packed words share.  TCAP_: tuples of a
// unit the emit phase takes (a unit of 150-base reads at k = 31, s = 11 has ~450).
template <int NW_, int MROWS_, int TCAP_, bool DMA_, int LIM>
struct SynPfLdsT {
    static constexpr int PR = 1, ROW = 33;  // (unused by the SEL machine; SynPk names them)
    static constexpr int NW = NW_;
    static constexpr bool DMA = DMA_;
    static constexpr int TCAP = TCAP_;
    static constexpr int TABK = 0, TABS = 0;  // the s-mer update table: 20 x uint4
    static constexpr int KTA = 320;           // u32x4 [64]: three bases, first half of a piece of six: (rol(F3, 3), R3)
    static constexpr int KT1 = KTA + 1024;    // u32x4 [4]: one base
    static constexpr bool KTB_DYN = DMA_;     // the short plan has no room for a second static table: KTB lies behind EBUF and is written at the start of every emit phase
    static constexpr int MROWS = MROWS_;
    static constexpr int PARK = KT1 + 64 + (KTB_DYN ? 0 : 1024);  // u32 [MROWS][64]
    static constexpr int MASK = PARK;
    static constexpr int EST = NW_;           // words from one lane's row of EBUF to the next (a window that runs over the row's end reads bits nobody uses)
    static constexpr int EBUF = PARK;         // u32 [64][EST]: the unit's packed words, in the emit phase
    static constexpr int KTB = KTB_DYN ? EBUF + 64 * EST * 4 : KT1 + 64;  // u32x4 [64]: second half of a piece of six: (F3, rol(R3, 3))
    static constexpr int SH = PARK, SP = PARK;
    static constexpr int FLAT = PARK + MROWS * 256;   // u16 [TCAP]: (lane << 9) | idx
    static constexpr int WBUF = FLAT + TCAP * 2;
    static constexpr int DBUF = WBUF + NW * 64 * 4;
    static constexpr int TOTAL = DMA ? DBUF + 512 : WBUF;
    static_assert(64 * EST * 4 + (KTB_DYN ? 1024 : 0) <= MROWS * 256 && WBUF % 16 == 0 && EBUF % 16 == 0 && TOTAL <= LIM && NW % 4 == 0, "SynPfLds");
};
typedef SynPfLdsT<PKNW, 20, 1024, true, 13648> SynPfLds;    // k_syncmer_pf: 13 184 B, twelve waves per CU, three per SIMD (<= 168 VGPRs)
typedef SynPfLdsT<32, 32, 1792, false, 20480> SynPfLdsL;     // k_syncmer_pfl: 14 208 B; two waves per SIMD (the registers of a 32-word prefetch), eight per CU

__device__ __forceinline__ void pf_rol64(u32 &lo, u32 &hi, u32 rot) {  // rot wave-uniform, 0..63
    if (rot & 32u) {
        const u32 t = lo;
        lo = hi;
        hi = t;
    }
    const u32 rs = rot & 31u;
    if (rs) {
        const u32 nl = __builtin_amdgcn_alignbit(lo, hi, 32u - rs), nh = __builtin_amdgcn_alignbit(hi, lo, 32u - rs);
        lo = nl;
        hi = nh;
    }
}

// canonical ntHash of the k bases from base idx of the read whose words are row `o` of EBUF -- from scratch.
// fwd = XOR_j rol(seed[b_j], k-1-j), rev = XOR_j rol(seed[comp b_j], j): XOR-linear in the bases, so any split of the k bases into
// consecutive pieces works.  A piece of SIX bases is two table rows -- KTA[c0] = (rol(F3, 3), R3) and KTB[c1] = (F3, rol(R3, 3)), c = three
// bases, 64 rows each -- and ONE rotation per strand: f = rol(f, 6) ^ A.f ^ B.f (Horner), and for rev the accumulator is kept rotated
// right by the number of bases taken so far, a = ror(a ^ A.r ^ B.r, 6), one rotation by k at the end puts it right.  Thirty bases (five
// sixes) come out of one 64-bit window of the read and their ten table reads are issued together; what is left of k (< 30 bases) goes
// six / three / one base at a time.  Bit-identical to the rolling form (the same 64-bit arithmetic in another order).
struct PfHash {
    u32 fl, fh, rl, rh;
};
__device__ __forceinline__ u32 pf_xor3(u32 a, u32 b, u32 c) {  // one full-rate v_bitop3_b32 (truth table 0x96 = parity) instead of two v_xor_b32
    u32 r;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// (q0, q1, q2: the three words of the read from word idx / 16 on -- the caller requests them a round ahead)
template <class LY>
__device__ __forceinline__ PfHash pf_hash_kmer(LDSQ const char *lds, u32 o, u32 idx, u32 k, u32 q0, u32 q1, u32 q2) {
    const LDSQ u32 *const row = reinterpret_cast<const LDSQ u32 *>(lds + LY::EBUF) + o * (u32)LY::EST;
    u32 fl = 0, fh = 0, al = 0, ah = 0;
    auto rows_of = [&](u32 offa, u32 offb, u32x4 &xa, u32x4 &xb) {
        xa = *reinterpret_cast<const LDSQ u32x4 *>(lds + LY::KTA + offa);
        xb = *reinterpret_cast<const LDSQ u32x4 *>(lds + LY::KTB + offb);
    };
    auto six = [&](const u32x4 &xa, const u32x4 &xb) {
        const u32 nfl = pf_xor3(__builtin_amdgcn_alignbit(fl, fh, 26), xa.x, xb.x), nfh = pf_xor3(__builtin_amdgcn_alignbit(fh, fl, 26), xa.y, xb.y);  // rol(f, 6)
        fl = nfl;
        fh = nfh;
        const u32 tl = pf_xor3(al, xa.z, xb.z), th = pf_xor3(ah, xa.w, xb.w);
        al = __builtin_amdgcn_alignbit(th, tl, 6);  // ror 6
        ah = __builtin_amdgcn_alignbit(tl, th, 6);
    };
    auto window = [&](u32 base, u32 &h0, u32 &h1) {  // 32 bases from base position `base` of the read on
        const LDSQ u32 *const wp = row + (base >> 4);
        const u32 q0 = wp[0], q1 = wp[1], q2 = wp[2], sh = (base & 15u) * 2u;
        h0 = __builtin_amdgcn_alignbit(q1, q0, sh);
        h1 = __builtin_amdgcn_alignbit(q2, q1, sh);
    };
    const u32 nseg = k / 30u;
    u32 rem = k - 30u * nseg;
    u32 h0 = 0, h1 = 0;
    {  // the first window: from the caller's words
        const u32 sh = (idx & 15u) * 2u;
        h0 = __builtin_amdgcn_alignbit(q1, q0, sh);
        h1 = __builtin_amdgcn_alignbit(q2, q1, sh);
    }
    auto one = [&](const u32x4 &x) {
        const u32 nfl = __builtin_amdgcn_alignbit(fl, fh, 31) ^ x.x, nfh = __builtin_amdgcn_alignbit(fh, fl, 31) ^ x.y;  // rol(f, 1)
        fl = nfl;
        fh = nfh;
        const u32 tl = al ^ x.z, th = ah ^ x.w;
        al = __builtin_amdgcn_alignbit(th, tl, 1);
        ah = __builtin_amdgcn_alignbit(tl, th, 1);
    };
    u32x4 xs0 = {0, 0, 0, 0}, xs1 = {0, 0, 0, 0};  // bases 30 and 31 of the last window: requested WITH its ten rows (k = 31 or 32, 61 or 62: no read of their own behind the chain)
    for (u32 sg = 0; sg < nseg; ++sg) {
        if (sg) window(idx + 30u * sg, h0, h1);
        const u32 mid = __builtin_amdgcn_alignbit(h1, h0, 24);  // bases 12..27
        u32x4 xa[5], xb[5];
        xs0 = *reinterpret_cast<const LDSQ u32x4 *>(lds + LY::KT1 + ((h1 >> 24) & 0x30u));
        xs1 = *reinterpret_cast<const LDSQ u32x4 *>(lds + LY::KT1 + ((h1 >> 26) & 0x30u));
        rows_of((h0 << 4) & 0x3f0u, (h0 >> 2) & 0x3f0u, xa[0], xb[0]);
        rows_of((h0 >> 8) & 0x3f0u, (h0 >> 14) & 0x3f0u, xa[1], xb[1]);
        rows_of((mid << 4) & 0x3f0u, (mid >> 2) & 0x3f0u, xa[2], xb[2]);
        rows_of((h1 >> 0) & 0x3f0u, (h1 >> 6) & 0x3f0u, xa[3], xb[3]);   // bases 18.. = bits 36..: h1 bits 4..
        rows_of((h1 >> 12) & 0x3f0u, (h1 >> 18) & 0x3f0u, xa[4], xb[4]);
#pragma unroll
        for (int q = 0; q < 5; ++q) six(xa[q], xb[q]);
    }
    if (nseg && rem && rem <= 2u) {  // (the last window still holds them)
        one(xs0);
        if (rem == 2u) one(xs1);
    } else if (rem) {
        u32 t0, t1;
        if (nseg) {
            window(idx + 30u * nseg, t0, t1);
        } else {
            t0 = h0;
            t1 = h1;
        }
        for (; rem >= 6u; rem -= 6u) {
            u32x4 xa, xb;
            rows_of((t0 << 4) & 0x3f0u, (t0 >> 2) & 0x3f0u, xa, xb);
            six(xa, xb);
            t0 = __builtin_amdgcn_alignbit(t1, t0, 12);
            t1 >>= 12;
        }
        if (rem >= 3u) {  // three bases: F3 is KTB's fwd half, R3 KTA's rev half
            const u32 off = (t0 << 4) & 0x3f0u;
            const u32x2 f3 = *reinterpret_cast<const LDSQ u32x2 *>(lds + LY::KTB + off);
            const u32x2 r3 = *reinterpret_cast<const LDSQ u32x2 *>(lds + LY::KTA + off + 8u);
            const u32 nfl = __builtin_amdgcn_alignbit(fl, fh, 29) ^ f3.x, nfh = __builtin_amdgcn_alignbit(fh, fl, 29) ^ f3.y;
            fl = nfl;
            fh = nfh;
            const u32 tl = al ^ r3.x, th = ah ^ r3.y;
            al = __builtin_amdgcn_alignbit(th, tl, 3);
            ah = __builtin_amdgcn_alignbit(tl, th, 3);
            t0 = __builtin_amdgcn_alignbit(t1, t0, 6);
            t1 >>= 6;
            rem -= 3u;
        }
        for (; rem; --rem) {
            const u32x4 x1 = *reinterpret_cast<const LDSQ u32x4 *>(lds + LY::KT1 + ((t0 << 4) & 0x30u));
            one(x1);
            t0 >>= 2;
        }
    }
    pf_rol64(al, ah, k & 63u);
    return PfHash{fl, fh, al, ah};
}

template <int W, class LY>
__device__ __forceinline__ void synpf_body(const KArgs &a, char *lds) {
    constexpr int NQ = LY::NW / 4;
    static_assert(W <= LY::MROWS, "synpf_body");
    LDSQ char *const ldsq = (LDSQ char *)lds;
    const int lane = lane_id();
    u32x4 ktb_row;  // this lane's row of KTB (the short plan keeps it in registers: the table's place is the hash phase's)
    {  // the tables, once per wavefront: s-mer update rows, the 64 three-base rows of KTA (and KTB where it has a place of its own), the four single bases
        SynPkTabs tabs;
        tabs.init(a.s, a.s, lane);
        if (lane < 20) *reinterpret_cast<LDSQ u32x4 *>(ldsq + LY::TABS + lane * 16) = tabs.row;
        const unsigned c0 = (unsigned)lane & 3u, c1 = ((unsigned)lane >> 2) & 3u, c2 = ((unsigned)lane >> 4) & 3u;  // c0 the FIRST base
        const u64 f3 = rol64(seed_fwd_code(c0), 2) ^ rol64(seed_fwd_code(c1), 1) ^ seed_fwd_code(c2);
        const u64 r3 = seed_rev_code(c0) ^ rol64(seed_rev_code(c1), 1) ^ rol64(seed_rev_code(c2), 2);
        const u64 fa = rol64(f3, 3), rb = rol64(r3, 3);
        *reinterpret_cast<LDSQ u32x4 *>(ldsq + LY::KTA + lane * 16) = (u32x4){(u32)fa, (u32)(fa >> 32), (u32)r3, (u32)(r3 >> 32)};
        ktb_row = (u32x4){(u32)f3, (u32)(f3 >> 32), (u32)rb, (u32)(rb >> 32)};
        if constexpr (!LY::KTB_DYN) *reinterpret_cast<LDSQ u32x4 *>(ldsq + LY::KTB + lane * 16) = ktb_row;
        if (lane < 4) {
            const u64 f1 = seed_fwd_code((unsigned)lane), r1 = seed_rev_code((unsigned)lane);
            *reinterpret_cast<LDSQ u32x4 *>(ldsq + LY::KT1 + lane * 16) = (u32x4){(u32)f1, (u32)(f1 >> 32), (u32)r1, (u32)(r1 >> 32)};
        }
        wave_sync_lds();
    }
    const u64 slab = (u64)64 * BSK_SYN_CAP;
    u64 d_cur = 0, d_nx = 0;
    SynWords<NQ == 4 || NQ == 6 || NQ == 8 ? NQ : 4> pw_cur;  // (register form of the prefetch)
#pragma unroll
    for (int j = 0; j < (int)(sizeof(pw_cur.q) / sizeof(pw_cur.q[0])); ++j) pw_cur.q[j] = (u32x4){0, 0, 0, 0};
    bool have = false;
    const u32 lseg = a.fixcap / a.list_grid;
    u32 lcur = 0;
    const u32 wbuf = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(size_t)(ldsq + LY::WBUF));
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
        const u64 rmax = a.n - 1;
        typename SynVec<LY::NW>::type wr;
        u64 d_n1, d_n2 = 0;
        SynWords<NQ == 4 || NQ == 6 || NQ == 8 ? NQ : 4> pw_n1;
        u32 rfl;
        if constexpr (LY::DMA) {
            if (!have) {  // first unit of a ticket: nothing was requested ahead
                d_cur = a.desc[r < rmax ? r : rmax];
                synpk_dma_words<NQ>(a.words + (d_cur >> 24), wbuf);
                synpk_dma_desc(a.desc + (r + 64 < rmax ? r + 64 : rmax), dbuf);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            {
                static_assert(!LY::DMA || NQ == 4, "synpf_body: the LDS-DMA plan takes 16 words");
                u32x4 wq[NQ];
                const LDSQ u32x4 *wb = reinterpret_cast<const LDSQ u32x4 *>(ldsq + LY::WBUF) + lane;
                const LDSQ u32 *db = reinterpret_cast<const LDSQ u32 *>(ldsq + LY::DBUF) + lane;
#pragma unroll
                for (int j = 0; j < NQ; ++j) wq[j] = wb[64 * j];
                u32 dl = db[0], dh = db[64];
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wq[0]), "+v"(wq[1]), "+v"(wq[2]), "+v"(wq[3]), "+v"(dl), "+v"(dh)::"memory");
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
            rfl = pk_load_u8(a.rflags ? a.rflags + (r < rmax ? r : rmax) : reinterpret_cast<const u8 *>(a.desc));
        } else {  // register form, as k_syncmer_pkl: every load unconditional, words of unit N+1 and descriptor of unit N+2 requested here
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
        const u64 ro = out_index(a, r, d);
        const long long Lorig = (long long)L - a.circ_ext;
        const bool ok = r < a.n && Lorig >= 0 && Lorig >= 2LL * a.k - a.s - 1 && L >= (u64)a.k;  // sketch.go:149
        const u32 nwin = ok ? (u32)(L - 2 * (u64)a.k + a.s + 2) : 0u;                             // end + 1
        const u32 ns = ok ? (u32)(L - a.s + 1) : 0u;
        const u32 ns_max = wave_max_u32(ns);
        const u32 nwin_min = ~wave_max_u32(ok ? ~nwin : 0u);
        u32 cnt = 0, tmin_lane = 0xffffffffu;
        if (ns_max) {
            SynPk<W, LY, 2> sp;
            sp.lds = ldsq;
            sp.k = a.k;
            sp.s = a.s;
            sp.lane = lane;
            sp.end_plus1 = nwin;
            sp.wr = wr;
            sp.gmask = nullptr;
            sp.run(ns_max, nwin_min, 0u, 0, 0u);
            cnt = sp.nsel;
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
        // reads the exact machine must run: two equal 27-bit keys met in one of their min operations (kernels_syncmer_pk.hpp) -- and the
        // last reads of a unit whose tuples do not fit the emit phase's list or the unit's slab (never on planned densities)
        u64 redo = __builtin_amdgcn_ballot_w64(ok && tmin_lane < 32u);
        if ((redo >> lane) & 1) cnt = 0;
        u32 incl = wave_incl_scan_u32(cnt, lane);
        const u64 over = __builtin_amdgcn_ballot_w64(incl > (u32)LY::TCAP);  // (a suffix of the lanes: incl never decreases)
        if (over) {
            redo |= over & __builtin_amdgcn_ballot_w64(cnt != 0u);
            if ((over >> lane) & 1) {
                cnt = 0;
                incl = 0;
            }
        }
        if (redo) list_append(a, reinterpret_cast<u32 *>(a.fixlist), lseg, lcur, redo, lane, r);
        const u32 excl = incl - cnt;
        const u32 T = wave_max_u32(incl);
        const u64 base = (u64)unit * slab;
#ifdef SYNPF_NOEXPAND  // dev knock-outs (timing only): the hash phase alone / + expansion / + hashes without their stores
        if (false) {
#else
        if (T) {
#endif
            // expand: tuple excl + j of the unit is (this lane, its j-th selected window); bit O of row mm: idx = (mm - 1) W + 1 + O
            const u32 nb = (ns_max + (u32)W - 1u) / (u32)W - 1u;
            LDSQ unsigned short *const flat = reinterpret_cast<LDSQ unsigned short *>(ldsq + LY::FLAT);
            {
                u32 at = ((u32)(size_t)flat) + excl * 2u;  // (byte address of this lane's next entry; the rows' bits are exactly its cnt selections: windows beyond a lane's end are masked in the hash phase)
                const u32 tag = (u32)lane << 9;
                for (u32 mm = 0; mm < nb; ++mm) {
                    u32 w = cnt ? *reinterpret_cast<const LDSQ u32 *>(ldsq + LY::MASK + mm * 256u + (u32)lane * 4u) : 0u;
                    const u32 val0 = tag + (mm * (u32)W + 1u - (u32)W);  // (+, not |: row 0 starts at idx 1 - W and only its last bit is ever set)
                    while (__builtin_amdgcn_ballot_w64(w != 0u)) {
                        if (w) {
                            const u32 O = (u32)__builtin_ctz(w);
                            w &= w - 1u;
                            *reinterpret_cast<LDSQ unsigned short *>(at) = (unsigned short)(val0 + O);
                            at += 2u;
                        }
                    }
                }
            }
            wave_sync_lds();  // every lane's rows are read: the words take their place
            {
                LDSQ u32x4 *eb = reinterpret_cast<LDSQ u32x4 *>(ldsq + LY::EBUF + lane * (LY::EST * 4));
#pragma unroll
                for (int j = 0; j < NQ; ++j) eb[j] = (u32x4){wr[4 * j], wr[4 * j + 1], wr[4 * j + 2], wr[4 * j + 3]};
                if constexpr (LY::KTB_DYN) *reinterpret_cast<LDSQ u32x4 *>(ldsq + LY::KTB + lane * 16) = ktb_row;
            }
            wave_sync_lds();
            u64 *const gh = a.hash + base;
            u32 *const gp = a.pos + base;
            // one lane per tuple, 64 per round; the round's list entry and the three words of its read are requested a round ahead
            auto request = [&](u32 tb, u32 &tp, u32 &q0, u32 &q1, u32 &q2) {
                const u32 tl = tb + (u32)lane;
                tp = (u32)flat[tl < T ? tl : T - 1u];
                const LDSQ u32 *const wp = reinterpret_cast<const LDSQ u32 *>(ldsq + LY::EBUF) + (tp >> 9) * (u32)LY::EST + ((tp & 0x1ffu) >> 4);
                q0 = wp[0];
                q1 = wp[1];
                q2 = wp[2];
            };
            u32 tp, q0, q1, q2;
            request(0u, tp, q0, q1, q2);
#ifdef SYNPF_NOEMIT
            for (u32 tb = 0; tb < 0; tb += 64) {
#else
            for (u32 tb = 0; tb < T; tb += 64) {
#endif
                const u32 tl = tb + (u32)lane;
                const bool live = tl < T;
                const u32 idx = tp & 0x1ffu, o = tp >> 9, c0 = q0, c1 = q1, c2 = q2;
                if (tb + 64u < T) request(tb + 64u, tp, q0, q1, q2);
                const PfHash h = pf_hash_kmer<LY>(ldsq, o, idx, (u32)a.k, c0, c1, c2);
                const bool rev = h.rh < h.fh || (h.rh == h.fh && h.rl < h.fl);  // nthash returns rev only when strictly smaller
#ifdef SYNPF_NOSTORE
                asm volatile("" ::"v"(h.rh), "v"(h.rl), "v"(h.fh), "v"(h.fl), "v"(rev), "v"(gh), "v"(gp));
#else
                if (live) {
                    __builtin_nontemporal_store(rev ? (((u64)h.rh << 32) | h.rl) : (((u64)h.fh << 32) | h.fl), &gh[tl]);
                    __builtin_nontemporal_store(idx | (rev ? BSK_POS_STRAND_BIT : 0u), &gp[tl]);
                }
#endif
            }
            wave_sync_lds();  // (the next unit's block 0 parks its suffix minima where EBUF is)
        }
        if (r < a.n && !((redo >> lane) & 1)) {
            a.refs[ro] = ((base + excl) << 24) | cnt;
            u8 sbyte = ok ? BSK_ST_OK : BSK_ST_SHORT;
            if (ok && a.rflags) sbyte |= (u8)rfl;
            a.status[ro] = sbyte;
        }
    }
    list_close(reinterpret_cast<u32 *>(a.fixlist), lseg, lcur, lane);
}

#ifndef SYNPF_LB
#define SYNPF_LB 3
#endif
template <int W>
__global__ __launch_bounds__(64, SYNPF_LB) void k_syncmer_pf(KArgs a) {  // three waves per SIMD: at most 168 VGPRs
    __shared__ __attribute__((aligned(16))) char lds[SynPfLds::TOTAL];
    synpf_body<W, SynPfLds>(a, lds);
}
// the same kernel for reads of up to 480 bases and k - s up to 24: 32 words of a read in registers, two waves per SIMD (k_syncmer_pkl's place)
template <int W>
__global__ __launch_bounds__(64, 2) void k_syncmer_pfl(KArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[SynPfLdsL::TOTAL];
    synpf_body<W, SynPfLdsL>(a, lds);
}

#ifndef BSK_SYNPF_WS
#define BSK_SYNPF_WS(X) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20)
#endif
#ifndef BSK_SYNPFL_WS
#define BSK_SYNPFL_WS(X) BSK_SYNPF_WS(X) X(21) X(22) X(23) X(24)
#endif
#ifdef BSK_IMPL_SYNPF
bool pf_syncmer_supported(int w, bool lng) {
#define X(WW) \
    if (w == WW) return true;
    if (lng) {
        BSK_SYNPFL_WS(X)
    } else {
        BSK_SYNPF_WS(X)
    }
#undef X
    return false;
}
u32 pf_syncmer_max_bases(bool lng) { return 16u * (u32)((lng ? SynPfLdsL::NW : SynPfLds::NW) - 2); }
u32 pf_syncmer_mask_rows(bool lng) { return (u32)(lng ? SynPfLdsL::MROWS : SynPfLds::MROWS); }
u32 pf_syncmer_unit_tuples(bool lng) { return (u32)(lng ? SynPfLdsL::TCAP : SynPfLds::TCAP); }
int pf_syncmer_blocks_per_cu(int w, bool lng) {
    int nb = 0;
    hipError_t e = hipErrorInvalidValue;
    if (lng) {
#define X(WW) \
    if (w == WW) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_syncmer_pfl<WW>, 64, 0);
        BSK_SYNPFL_WS(X)
#undef X
    } else {
#define X(WW) \
    if (w == WW) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_syncmer_pf<WW>, 64, 0);
        BSK_SYNPF_WS(X)
#undef X
    }
    if (e != hipSuccess || nb < 1) {
        (void)hipGetLastError();
        nb = 1;
    }
    return nb;
}
// the fused kernel, then the exact machine over the listed reads (k_syncmer_fix.hip)
void pf_syncmer_launch(int w, bool lng, int grid, int fix_grid, hipStream_t stream, const KArgs &a) {
    if (lng) {
#define X(WW) \
    if (w == WW) hipLaunchKernelGGL((k_syncmer_pfl<WW>), dim3(grid), dim3(64), 0, stream, a);
        BSK_SYNPFL_WS(X)
#undef X
    } else {
#define X(WW) \
    if (w == WW) hipLaunchKernelGGL((k_syncmer_pf<WW>), dim3(grid), dim3(64), 0, stream, a);
        BSK_SYNPF_WS(X)
#undef X
    }
    pk_syncmer_fix_launch(w, fix_grid, stream, a);
}
#endif  // BSK_IMPL_SYNPF

}  // namespace bsk
