// kernels_protein.hpp -- protein minimizer (NewProteinMinimizerSketch / Next,
// sketch-protein.go:62-210) specialised on the window W and the k-mer size K (9..16).
//
// One sequence per lane.  wyhash (seed 1) of every K residues is computed from scratch
// per position as the reference does (sketch-protein.go:117), but entirely in registers:
// the lane keeps the next 20 residues in five dwords; at sub-step j the K bytes start at
// byte j, so r32(p), r32(p+4) and the tail are three v_alignbit with an immediate shift;
// the two 64x64->128 "mum" products are four v_mad_u64_u32 each.  The leftmost-argmin
// window machine is the one of kernels_fast.hpp.
//
// Output: a 300-residue sequence selects ~97 positions, far more than can be staged per
// lane for a whole sequence, so every sequence owns a fixed slab of `slab_read` tuples in
// the tuple arrays and the wavefront flushes its LDS staging every MB = lcm(W,4) steps
// (each lane appends its new tuples to its own slab; runs leave LDS in lane order).
#pragma once
#include "kernels_fast.hpp"
#include "kernels_more.hpp"
#include "kernels_translate.hpp"

namespace bsk {

typedef u32 u32_u __attribute__((aligned(1)));

constexpr int ilcm4(int w) { return (w % 4 == 0) ? w : (w % 2 == 0 ? 2 * w : 4 * w); }

template <int MB, int G = 16>
struct ProtLds {
    static constexpr int ROW = 65;
    static constexpr int ROWS = MB + G;                       // the left-over of a flush group (< G) + MB new tuples + 1 spare row
    static constexpr int SH = 0;                              // u64 [ROWS*65]
    static constexpr int SP = SH + ROWS * ROW * 8;            // u16 [ROWS*65]
    static constexpr int EXCL = SP + ((ROWS * ROW * 2 + 15) & ~15);
    static constexpr int HEADS = EXCL + 256;                  // u64 [ROWS]
    static constexpr int NZ = HEADS + ROWS * 8;
    static constexpr int DST = NZ + 64;                       // u32 [64]: lane's next free tuple index inside the unit's region, u32 [64]: ring heads
    static constexpr int TOTAL = DST + 512;
};

__device__ __forceinline__ u64 mum64(u32 a0, u32 a1, u32 b0, u32 b1) {  // hi64(a*b) ^ lo64(a*b)
    const u64 t0 = (u64)a0 * b0;
    const u64 t1 = (u64)a1 * b0 + (t0 >> 32);
    const u64 t2 = (u64)a0 * b1 + (u32)t1;
    const u64 hi = (u64)a1 * b1 + (t1 >> 32) + (t2 >> 32);
    const u64 lo = (t2 << 32) | (u32)t0;
    return hi ^ lo;
}

#ifndef PROT_WIDE
#define PROT_WIDE 1  // the residues of four macro blocks per request (k_prot_minimizer_fast; 0: one block per request, round 3's way)
#endif
template <int W, int K>
struct FastProt {
    static_assert(K >= 4 && K <= 16, "register wyhash covers 4..16 residues");
    static constexpr int MB = ilcm4(W);  // steps per macro block: a whole number of windows-blocks and of dwords
    // Tuples leave in whole groups of 16 (full 128-byte lines of hashes; groups of 8 were 18 % faster through two more waves per
    // CU but doubled the HBM traffic: half-line writes are read-modify-write).  The staging holds the left-over of a group
    // (< 16) plus what the steps between two flushes can select (HS), so where a whole macro block does not leave 8 waves per
    // CU the flush also runs in the middle of it.
    static constexpr int GL = 4, G = 16;
    static constexpr int HS = (ProtLds<MB, 16>::TOTAL <= 20480) ? MB : MB / 2;
    typedef ProtLds<HS, G> LY;
    LDSQ char *lds;
    int lane;
    u32 nk;
    HV S[W], P;
    u32 prev, slot, send;  // send: the slot offset one row past the lane's last row
    u32 R[5 + MB / 4];  // residues: R[0..4] = the 20 bytes at the macro block's first position, R[5..] = the following ones
    lmask tm;

    // wyhash (published v1) of the K bytes starting at byte j of (A,B,C,D,E)
    template <int J>
    __device__ __forceinline__ u64 hashK(u32 A, u32 B, u32 C, u32 D, u32 E) const {
        const u32 h0 = J ? __builtin_amdgcn_alignbit(B, A, 8 * J) : A;  // r32(p)
        const u32 h1 = J ? __builtin_amdgcn_alignbit(C, B, 8 * J) : B;  // r32(p+4)
        if constexpr (K <= 8) {  // (round 5) 4..8 residues: one product with the whole k-mer as its tail, wymum(seed, tail ^ p1)
            u64 x;
            if (K == 4) x = h0;
            else if (K == 5) x = ((u64)h0 << 8) | (h1 & 0xffu);
            else if (K == 6) x = ((u64)h0 << 16) | (h1 & 0xffffu);
            else if (K == 7) x = ((u64)h0 << 24) | ((u64)(h1 & 0xffffu) << 8) | ((h1 >> 16) & 0xffu);
            else x = ((u64)h0 << 32) | h1;
            constexpr u64 sd = 1ULL ^ WYP0;
            const u64 bb = x ^ WYP1;
            const u64 s1 = mum64((u32)sd, (u32)(sd >> 32), (u32)bb, (u32)(bb >> 32));
            constexpr u64 fin = (u64)K ^ WYP5;
            return mum64((u32)s1, (u32)(s1 >> 32), (u32)fin, (u32)(fin >> 32));
        }
        const u32 t0 = J ? __builtin_amdgcn_alignbit(D, C, 8 * J) : C;  // bytes 8..11
        const u32 t1 = (K > 12) ? (J ? __builtin_amdgcn_alignbit(E, D, 8 * J) : D) : 0u;  // bytes 12..15
        constexpr int r = K - 8;
        u64 tail;
        if (r == 1) tail = t0 & 0xffu;
        else if (r == 2) tail = t0 & 0xffffu;
        else if (r == 3) tail = ((t0 & 0xffffu) << 8) | ((t0 >> 16) & 0xffu);
        else if (r == 4) tail = t0;
        else if (r == 5) tail = ((u64)t0 << 8) | (t1 & 0xffu);
        else if (r == 6) tail = ((u64)t0 << 16) | (t1 & 0xffffu);
        else if (r == 7) tail = ((u64)t0 << 24) | ((u64)(t1 & 0xffffu) << 8) | ((t1 >> 16) & 0xffu);
        else tail = ((u64)t0 << 32) | t1;
        constexpr u64 seed0 = 1ULL ^ WYP0;                  // seed ^= p0
        const u64 a = (((u64)h0 << 32) | h1) ^ seed0;       // wyr64s(p) ^ seed
        const u64 b = tail ^ WYP2;
        const u64 s1 = mum64((u32)a, (u32)(a >> 32), (u32)b, (u32)(b >> 32));
        constexpr u64 fin = (u64)K ^ WYP5;
        return mum64((u32)s1, (u32)(s1 >> 32), (u32)fin, (u32)(fin >> 32));
    }

    // steps T0 .. T1-1 of the macro block (MB steps) starting at k-mer position i0 (FIRSTMB: i0 == 0)
    template <bool FIRSTMB, int T0, int T1>
    __device__ __forceinline__ void macro(u32 i0) {
        u32 vi = i0 + T0;
#pragma unroll
        for (int t = T0; t < T1; ++t) {
            const int g = t >> 2;
            u64 h;
            switch (t & 3) {
                case 0: h = hashK<0>(R[g], R[g + 1], R[g + 2], R[g + 3], R[g + 4]); break;
                case 1: h = hashK<1>(R[g], R[g + 1], R[g + 2], R[g + 3], R[g + 4]); break;
                case 2: h = hashK<2>(R[g], R[g + 1], R[g + 2], R[g + 3], R[g + 4]); break;
                default: h = hashK<3>(R[g], R[g + 1], R[g + 2], R[g + 3], R[g + 4]); break;
            }
            const int o = t % W;
            const bool first = FIRSTMB && t < W;  // the very first window block of the sequence
            HV v;
            v.lo = (u32)h;
            v.hi = (u32)(h >> 32);
            v.p = vi;
            if (o == 0) P = v;
            else P = selv(lt64(v.lo, v.hi, P.lo, P.hi), v, P);
            if (!first || o == W - 1) {
                HV m = P;
                if (o != W - 1) {
                    const int o1 = o + 1 < W ? o + 1 : o;  // (folds when the loop is unrolled; keeps the dead last iteration inside the array)
                    m = selv(lt64(P.lo, P.hi, S[o1].lo, S[o1].hi), P, S[o1]);
                }
                const lmask e = __builtin_amdgcn_ballot_w64(m.p != prev) & __builtin_amdgcn_ballot_w64(vi < nk);
                prev = m.p;
                *reinterpret_cast<LDSQ u64 *>(lds + LY::SH + slot) = ((u64)m.hi << 32) | m.lo;
                *reinterpret_cast<LDSQ u16 *>(lds + LY::SP + (slot >> 2)) = (u16)m.p;
                u32 nxt = slot + (u32)(LY::ROW * 8);  // the lane's rows are a ring: past the last row comes row 0
                nxt = sel(__builtin_amdgcn_ballot_w64(nxt == send), (u32)lane * 8u, nxt);
                slot = sel(e, nxt, slot);
            }
            S[o] = v;
            vi += 1;
            if (o == W - 1) {
                if (first) suffix_min_pass<W, true>(S, tm);  // + BSK_ST_FIRST_WINDOW_TIE (kernels_fast.hpp)
                else suffix_min_pass<W, false>(S, tm);
            }
        }
    }
};

// DNA: the batch is 2-bit packed DNA and the residues are translated where they are fetched (Translate with the constructor's
// arguments, sketch-protein.go:83-88): a dword of four residues is 12 bases = one 24-bit field of the packed stream, and a
// codon's amino acid is one byte of a 64-entry LDS table indexed by the codon's raw 6 bits (one table for the plus frames, one
// with the reverse-complemented codons for the minus frames, which walk the sequence downwards).  ~5 instructions per residue
// on top of ~60 for its hash and window; no translated copy of the batch ever exists.
typedef u64 u64_a4 __attribute__((aligned(4)));
// Residues of a 2-bit DNA sequence, four at a time, translated where they are fetched (the fused DNA-fed protein kernels).
// Residue dword jd = residues 4jd .. 4jd+3 = 12 bases = one 24-bit field F of the packed stream, lowest base `lo`.
// Plus frames: residue t is F's bits [6t, 6t+6).  Minus frames walk downwards: residue t's codon ends 3t bases below the
// field's top base Lnt + frame - 12 jd, so it is bits [18-6t, 24-6t) (complemented and reversed by the table); a field
// that starts before base 0 (the sequence's last residues) is shifted up instead of moved.
// aat: 64-entry LDS table, amino acid of the codon whose three bases, lowest position first, are the 2-bit codes of the index.
struct DnaResidues {
    const u32 *wseq;
    u32 Lnt;
    int frame;
    const LDSQ u8 *aat;
    static __device__ __forceinline__ void build_table(u8 *tab, const u8 *lut, int frame, int lane) {
        const u32 six = (u32)lane, b0 = six & 3, b1 = (six >> 2) & 3, b2 = six >> 4;
        const u32 idx = frame > 0 ? (b0 << 4) | (b1 << 2) | b2 : ((b2 ^ 3) << 4) | ((b1 ^ 3) << 2) | (b0 ^ 3);
        tab[lane] = lut[4096 + 256 + idx];
    }
    __device__ __forceinline__ int lo_of(u32 jd) const { return frame > 0 ? (frame - 1) + 12 * (int)jd : (int)Lnt + frame - 12 * (int)jd - 11; }
    __device__ __forceinline__ u64 issue(u32 jd) const {  // the two packed words that hold the field (request them early)
        int lo = lo_of(jd);
        lo = lo < 0 ? 0 : (lo > (int)Lnt ? (int)Lnt : lo);  // (fields wholly past the end are never hashed; stay inside the buffer)
        return *reinterpret_cast<const GLBQ u64_a4 *>((size_t)(wseq + ((u32)lo >> 4)));
    }
    __device__ __forceinline__ u32 finish(u64 two, u32 jd) const {
        const int lo = lo_of(jd);
        u32 F;
        if (lo < 0) F = (u32)(two << (2u * (u32)(lo < -12 ? 12 : -lo)));
        else F = (u32)(two >> (((u32)(lo > (int)Lnt ? (int)Lnt : lo) & 15u) * 2u));
        if (frame > 0) return (u32)aat[F & 63u] | ((u32)aat[(F >> 6) & 63u] << 8) | ((u32)aat[(F >> 12) & 63u] << 16) | ((u32)aat[(F >> 18) & 63u] << 24);
        return (u32)aat[(F >> 18) & 63u] | ((u32)aat[(F >> 12) & 63u] << 8) | ((u32)aat[(F >> 6) & 63u] << 16) | ((u32)aat[F & 63u] << 24);
    }
    // Plus frames, four dwords at once: dwords jd0 .. jd0+3 (jd0 a multiple of 4) are 48 bases = three packed words starting at word
    // 3 jd0 / 4, shifted by the frame's 0 / 2 / 4 bits -- the same wave-uniform alignment for every jd0, so ONE 16-byte load serves the
    // four fields and each field is one v_alignbit with a scalar shift (no per-field address arithmetic, a quarter of the loads).
    // (Reads up to maxlen / 16 + 4 words past a shorter sequence's end: inside the batch's slack, the residues are never stored.)
    __device__ __forceinline__ u32x4 issue4(u32 jd0) const { return *reinterpret_cast<const GLBQ u32x4_u *>((size_t)(wseq + 3u * (jd0 >> 2))); }
    __device__ __forceinline__ void finish4(const u32x4 &w4, u32 (&out)[4]) const {
        const u32 d = 2u * (u32)(frame - 1);
        const u32 F[4] = {__builtin_amdgcn_alignbit(w4.y, w4.x, d), __builtin_amdgcn_alignbit(w4.y, w4.x, 24u + d),
                          __builtin_amdgcn_alignbit(w4.z, w4.y, 16u + d), __builtin_amdgcn_alignbit(w4.w, w4.z, 8u + d)};
#pragma unroll
        for (int g = 0; g < 4; ++g)
            out[g] = (u32)aat[F[g] & 63u] | ((u32)aat[(F[g] >> 6) & 63u] << 8) | ((u32)aat[(F[g] >> 12) & 63u] << 16) | ((u32)aat[(F[g] >> 18) & 63u] << 24);
    }
    // Minus frames, four dwords at once: the fields of dwords jd0 .. jd0+3 descend from the sequence's END, so their alignment is the
    // lane's own -- but it is the same for every jd0 (48 bases = three words per step): with A = lowest base of the four fields,
    // a = A & 15 is a per-lane constant, field g sits 2a + 24(3-g) bits into the window of four words at word A >> 4, i.e. in one of
    // two word pairs chosen by a (three per-lane masks) and at a per-lane shift.  Near the sequence's start A is negative: the window
    // then begins in the PREVIOUS sequence's words, whose bits only reach residues past the end of this translation (never stored);
    // only the batch's very first sequence has nothing before it (first_word + A >> 4 < 0): such a lane says so and takes finish().
    __device__ __forceinline__ int low_base4(u32 jd0) const { return (int)Lnt + frame - 12 * (int)(jd0 + 3) - 11; }
    __device__ __forceinline__ u32x4 issue4m(u32 jd0, long long first_word, bool &before_batch) const {
        const int A = low_base4(jd0);
        long long w0 = (long long)(A >> 4);  // floor
        before_batch = first_word + w0 < 0;
        if (before_batch) w0 = -first_word;
        return *reinterpret_cast<const GLBQ u32x4_u *>((size_t)(wseq + w0));
    }
    __device__ __forceinline__ void finish4m(const u32x4 &w4, u32 jd0, u32 (&out)[4]) const {
        const u32 a = (u32)low_base4(jd0) & 15u;
        const u32 s3 = 2u * a, s2 = (24u + 2u * a) & 31u, s1 = (48u + 2u * a) & 31u, s0 = (72u + 2u * a) & 31u;
        const bool u2 = a >= 4u, u1 = a >= 8u, u0 = a >= 12u;  // the field starts one word higher
        const u32 F[4] = {__builtin_amdgcn_alignbit(u0 ? w4.w : w4.w, u0 ? w4.w : w4.z, s0),  // (u0: the field lies inside word 3: bits 8..)
                          __builtin_amdgcn_alignbit(u1 ? w4.w : w4.z, u1 ? w4.z : w4.y, s1),
                          __builtin_amdgcn_alignbit(u2 ? w4.z : w4.y, u2 ? w4.y : w4.x, s2),
                          __builtin_amdgcn_alignbit(w4.y, w4.x, s3)};
#pragma unroll
        for (int g = 0; g < 4; ++g)
            out[g] = (u32)aat[(F[g] >> 18) & 63u] | ((u32)aat[(F[g] >> 12) & 63u] << 8) | ((u32)aat[(F[g] >> 6) & 63u] << 16) | ((u32)aat[F[g] & 63u] << 24);
    }
};

template <int W, int K, bool DNA = false>
__global__ __launch_bounds__(64) void k_prot_minimizer_fast(KArgs a) {
    typedef FastProt<W, K> FP;
    constexpr int MB = FP::MB, GL = FP::GL, G = FP::G, HS = FP::HS;
    typedef typename FP::LY LY;
    __shared__ __attribute__((aligned(16))) char lds[LY::TOTAL + (DNA ? 64 : 0)];
    LDSQ char *const ldsq = (LDSQ char *)lds;
    const int lane = lane_id();
    const int frame = a.frame;
    if (DNA) {
        DnaResidues::build_table(reinterpret_cast<u8 *>(lds + LY::TOTAL), a.lut, frame, lane);
        __syncthreads();
    }
    const u64 slab_read = a.slab_read;  // tuples reserved per sequence
    const u32 tku = a.tk ? a.tk : 4u;  // (KArgs::tk: fewer for small batches)
    for (u32 unit = next_ticket(a.ticket, lane) * tku, uend = unit + tku; unit < a.nunits; ++unit, ({
             if (unit == uend) {
                 unit = next_ticket(a.ticket, lane) * tku;
                 uend = unit + tku;
             }
         })) {
        const u64 r = (u64)unit * 64 + lane;
        u64 off = 0, L = 0;
        u32 Lnt = 0;  // DNA: nucleotides of the sequence
        bool ok;
        if (DNA) {
            if (r < a.n) {
                const u64 d = a.desc[r];
                off = d >> 24;
                Lnt = (u32)(d & 0xffffffULL);
            }
            ok = r < a.n && (u64)Lnt >= (u64)K * 3 + (u64)W - 1;  // sketch-protein.go:66,73: on the nucleotide length
            L = translated_len((u64)Lnt, frame);                   // codon_tables.go:224,256
        } else {
            if (r < a.n) ascii_span(a, r, off, L);
            ok = r < a.n && prot_len_ok(a, r, L, (u64)K * 3 + (u64)W - 1);  // sketch-protein.go:66,73
        }
        const u32 nk = (ok && L >= (u64)K + (u64)W - 1) ? (u32)(L - K + 1) : 0u;
        const u32 nk_max = wave_max_u32(nk);
        u32 done = 0;  // tuples of this lane already in HBM
        u32 tie = 0;
        const u64 ubase = (u64)unit * 64 * slab_read;
        if (nk_max) {
            FP fp;
            fp.lds = ldsq;
            fp.lane = lane;
            fp.nk = nk;
            fp.prev = 0xffffffffu;
            fp.tm = 0;
            // residues are bytes at an arbitrary address: byte-aligned 16- and 4-byte global loads (unaligned access mode of
            // the HSA ABI).  Five aligned dword loads + realignment per macro block touched every cache line 32 times; with
            // 6 wavefronts per CU walking 64 sequences each, the 48 KB of lines do not stay in the 32 KB L1.
            const u8 *p0 = a.ascii + off;
            const size_t gp0 = (size_t)p0;
            const DnaResidues dr{a.words + off, Lnt, frame, reinterpret_cast<const LDSQ u8 *>(ldsq + LY::TOTAL)};  // (DNA only)
            auto dna_issue = [&](u32 jd) -> u64 { return dr.issue(jd); };
            auto dna_finish = [&](u64 two, u32 jd) -> u32 { return dr.finish(two, jd); };
            auto load_dwords = [&](u32 *dst, int ndw, u32 j) {  // dwords j .. j+ndw-1 of the sequence (bytes 4j ..)
                int g = 0;
                if (DNA) {
#pragma unroll
                    for (; g < ndw; ++g) dst[g] = dna_finish(dna_issue(j + (u32)g), j + (u32)g);
                    return;
                }
                for (; g + 4 <= ndw; g += 4) {
                    const u64 bo = (u64)4 * (j + g) < L ? (u64)4 * (j + g) : L;  // never start beyond the sequence (+ buffer slack)
                    const u32x4 v = *reinterpret_cast<const GLBQ u32x4_u *>(gp0 + bo);
                    dst[g] = v.x;
                    dst[g + 1] = v.y;
                    dst[g + 2] = v.z;
                    dst[g + 3] = v.w;
                }
                for (; g < ndw; ++g) {
                    const u64 bo = (u64)4 * (j + g) < L ? (u64)4 * (j + g) : L;
                    dst[g] = *reinterpret_cast<const GLBQ u32_u *>(gp0 + bo);
                }
            };
            u32 dj = 0;
            fp.slot = (u32)lane * 8u;  // staging persists across macro blocks (leftovers of less than a flush group stay in LDS)
            fp.send = (u32)(LY::ROWS * LY::ROW + lane) * 8u;
            load_dwords(fp.R, 5 + MB / 4, 0);
            dj = 5 + MB / 4;
            // ---- flush whole groups of 16 tuples (full 128-byte lines of hashes) of every lane to its slab ----
            // The lane's ROWS rows are a ring (head = row of its oldest staged tuple): moving the < 16 left-over tuples down to
            // row 0 after every flush was a third of the kernel's instructions.
            u32 head = 0;
            auto flush = [&](bool last) {  // last: everything that is staged
                const u32 wrow = (fp.slot - (u32)lane * 8u) / (u32)(LY::ROW * 8);  // row of the next write
                const u32 cnt = wrow >= head ? wrow - head : wrow + (u32)LY::ROWS - head;  // staged (< ROWS: left-over < 16, new <= HS)
                if (last) flush_last<LY, false, GL, LY::ROWS>(lds, lane, cnt, done, slab_read, ubase, a, head);
                else flush_groups<LY, false, GL, LY::ROWS>(lds, lane, cnt, done, slab_read, ubase, a, head);
                const u32 nfl = last ? cnt : (cnt & ~(u32)(G - 1));
                head += nfl;
                head = head >= (u32)LY::ROWS ? head - (u32)LY::ROWS : head;
                done += nfl;
            };
            // PROT_WIDE (protein-fed): the residues of FOUR macro blocks are requested together, every fourth block -- a 128-byte line is touched
            // two or three times instead of eight, and its touches lie four blocks apart at most (the 16-byte requests of every block keep
            // 64 lanes' partly consumed lines alive for eight blocks: at eight waves per CU they outgrow the L2 and are fetched 1.5 times).
            constexpr int QB = PROT_WIDE ? 4 : 1, ND = MB / 4;
            u32 wide[PROT_WIDE ? QB * ND : 1];
            auto one_block = [&](auto qc, const u32 i0) {
                constexpr int Q = decltype(qc)::value;
                u64 raw[MB / 4];  // DNA: the next macro block's packed words are requested here, a whole block of hashing ahead of their use
                u32 nextR[MB / 4];  // protein: the same for the next macro block's residues (round 3: they were requested after the
                                    // block and waited for at once -- an exposed L2 / HBM round trip every MB steps)
                if (DNA) {
#pragma unroll
                    for (int g = 0; g < MB / 4; ++g) raw[g] = dna_issue(dj + (u32)g);
                } else {
#ifndef PROT_NO_AHEAD
                    if (PROT_WIDE && !DNA) {
                        if (Q == 0) load_dwords(wide, QB * ND, dj);
                    } else {
                        load_dwords(nextR, MB / 4, dj);
                    }
#endif
                }
                if (HS < MB) {
                    if (i0 == 0) fp.template macro<true, 0, HS>(i0);
                    else fp.template macro<false, 0, HS>(i0);
                    flush(i0 + HS >= nk_max);
                    if (i0 == 0) fp.template macro<true, HS, MB>(i0);
                    else fp.template macro<false, HS, MB>(i0);
                } else {
                    if (i0 == 0) fp.template macro<true, 0, MB>(i0);
                    else fp.template macro<false, 0, MB>(i0);
                }
#pragma unroll
                for (int g = 0; g < 5; ++g) fp.R[g] = fp.R[g + MB / 4];
                // residues of the NEXT macro block: requested and waited for BEFORE this round's flush stores are
                // issued (vmcnt is in-order: a load issued after the stores could only be waited for together with them)
                if (DNA) {
#pragma unroll
                    for (int g = 0; g < MB / 4; ++g) fp.R[5 + g] = dna_finish(raw[g], dj + (u32)g);
                } else {
#ifndef PROT_NO_AHEAD
                    if (PROT_WIDE && !DNA) {
                        if (Q == 0) {
#pragma unroll
                            for (int g = 0; g < QB * ND; ++g) asm volatile("" : "+v"(wide[g]));  // (all of them in before the flush's stores go out)
                        }
#pragma unroll
                        for (int g = 0; g < ND; ++g) fp.R[5 + g] = wide[(Q % QB) * ND + g];
                    } else {
#pragma unroll
                        for (int g = 0; g < MB / 4; ++g) fp.R[5 + g] = nextR[g];
                    }
#else
                    load_dwords(fp.R + 5, MB / 4, dj);
#endif
                }
                dj += MB / 4;
#pragma unroll
                for (int g = 0; g < MB / 4; ++g) asm volatile("" ::"v"(fp.R[5 + g]));
                flush(i0 + MB >= nk_max);
            };
            {
                u32 i0 = 0;
                while (i0 < nk_max) {
                    one_block(std::integral_constant<int, 0>{}, i0);
                    i0 += MB;
                    if (PROT_WIDE && !DNA) {
                        if (i0 >= nk_max) break;
                        one_block(std::integral_constant<int, 1>{}, i0);
                        i0 += MB;
                        if (i0 >= nk_max) break;
                        one_block(std::integral_constant<int, 2>{}, i0);
                        i0 += MB;
                        if (i0 >= nk_max) break;
                        one_block(std::integral_constant<int, 3>{}, i0);
                        i0 += MB;
                    }
                }
            }
            tie = (u32)((fp.tm >> lane) & 1);
        }
        if (r < a.n) {
            a.refs[r] = ((ubase + (u64)lane * slab_read) << 24) | done;
            a.status[r] = (u8)((ok ? BSK_ST_OK : BSK_ST_SHORT) | ((ok && tie) ? BSK_ST_FIRST_WINDOW_TIE : 0));
        }
    }
}

// every window 2..8 with every k 4..16 (the register wyhash covers 4..16 residues), protein-fed and DNA-fed, in two translation units:
// k_protein.hip k = 9..16, k_protein_short.hip k = 4..8 (round 5: k = 7 w = 3 ran on the general kernel, 30 G residues/s against 550)
bool fast_prot_short_supported(int w, int k);
int fast_prot_short_blocks_per_cu(int w, int k);
void fast_prot_short_launch(int w, int k, int grid, hipStream_t stream, const KArgs &a, bool dna);
#if defined(BSK_IMPL_PROTEIN) || defined(BSK_IMPL_PROTEIN_SHORT)
#ifdef BSK_IMPL_PROTEIN_SHORT
#define BSK_PROT_K(X, WW) X(WW, 4) X(WW, 5) X(WW, 6) X(WW, 7) X(WW, 8)
#define BSK_PROT_FN(name) fast_prot_short_##name
#else
#ifndef BSK_PROT_KW  // (dev builds narrow the list: -D'BSK_PROT_KW(X)=X(5,9)')
#define BSK_PROT_K(X, WW) X(WW, 9) X(WW, 10) X(WW, 11) X(WW, 12) X(WW, 13) X(WW, 14) X(WW, 15) X(WW, 16)
#endif
#define BSK_PROT_FN(name) fast_prot_long_##name
#endif
#ifndef BSK_PROT_KW
#define BSK_PROT_KW(X) BSK_PROT_K(X, 2) BSK_PROT_K(X, 3) BSK_PROT_K(X, 4) BSK_PROT_K(X, 5) BSK_PROT_K(X, 6) BSK_PROT_K(X, 7) BSK_PROT_K(X, 8)
#endif
bool BSK_PROT_FN(supported)(int w, int k) {
#define X(WW, KK) \
    if (w == WW && k == KK) return true;
    BSK_PROT_KW(X)
#undef X
    return false;
}
int BSK_PROT_FN(blocks_per_cu)(int w, int k) {
    int nb = 0;
    hipError_t e = hipErrorInvalidValue;
#define X(WW, KK) \
    if (w == WW && k == KK) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_prot_minimizer_fast<WW, KK, false>, 64, 0);
    BSK_PROT_KW(X)
#undef X
    if (e != hipSuccess || nb < 1) {
        (void)hipGetLastError();
        nb = 1;
    }
    return nb;
}
void BSK_PROT_FN(launch)(int w, int k, int grid, hipStream_t stream, const KArgs &a, bool dna) {  // dna: 2-bit DNA in, translated on the fly
#define X(WW, KK)                                                                                                              \
    if (w == WW && k == KK) {                                                                                                  \
        if (dna) hipLaunchKernelGGL((k_prot_minimizer_fast<WW, KK, true>), dim3(grid), dim3(64), 0, stream, a);               \
        else hipLaunchKernelGGL((k_prot_minimizer_fast<WW, KK, false>), dim3(grid), dim3(64), 0, stream, a);                  \
    }
    BSK_PROT_KW(X)
#undef X
}
#endif
#ifdef BSK_IMPL_PROTEIN
bool fast_prot_supported(int w, int k) { return k <= 8 ? fast_prot_short_supported(w, k) : fast_prot_long_supported(w, k); }
int fast_prot_blocks_per_cu(int w, int k) { return k <= 8 ? fast_prot_short_blocks_per_cu(w, k) : fast_prot_long_blocks_per_cu(w, k); }
void fast_prot_launch(int w, int k, int grid, hipStream_t stream, const KArgs &a) {
    if (k <= 8) fast_prot_short_launch(w, k, grid, stream, a, false);
    else fast_prot_long_launch(w, k, grid, stream, a, false);
}
void fast_prot_dna_launch(int w, int k, int grid, hipStream_t stream, const KArgs &a) {
    if (k <= 8) fast_prot_short_launch(w, k, grid, stream, a, true);
    else fast_prot_long_launch(w, k, grid, stream, a, true);
}
#endif  // BSK_IMPL_PROTEIN

// ---------------------------------------------------------------------------------------
// Protein k-mer hashes (kind BSK_PROT_HASH; ProteinIterator.Next, iterator-protein.go:86-117)
// for K = 9..16: value i of sequence r -> hash[first(r) + i].  One sequence per lane; the
// residues of a 256-position chunk are staged in LDS (one region per lane, cooperative coalesced loads) so that
// the hashing loop has no global loads (see k_nthash_fast: loads and stores in one loop make
// every block wait for the previous flush); 16 hashes per lane go through the 64x16 tile and
// leave as one aligned 128-byte line per sequence.  Runs are padded to 16 values.
// ---------------------------------------------------------------------------------------
#ifndef BSK_PH_CHUNK
#define BSK_PH_CHUNK 256
#endif

// Cooperative staging of one residue chunk per sequence: for every source lane s in `want`, the wavefront copies
// np16 16-byte pieces starting at that lane's (byte-aligned) address A into the lane's LDS region (RS dwords apart,
// RS = 4*odd: regions are 16-byte aligned and ds_read_b128 of 64 different regions is 2-way conflicted at most).
// One global_load_dwordx4 per source touches ~np16/8 lines (a per-lane walk touches 64 lines per instruction and
// measured 2x slower end to end: the 64 KB of lines that 8 waves walk do not stay in the 32 KB L1).
#ifndef BSK_PH_UNR
#define BSK_PH_UNR 16
#endif
// Branch-free on purpose: UNR loads are issued back to back and every lane takes part (lanes past the last piece
// repeat it) -- with a branch around each load the compiler waits for vmcnt(0) between them and the 64 loads of a
// chunk run one memory latency after the other.
template <int RS, int NP>
__device__ __forceinline__ void stage_chunks(LDSQ char *reg0, const u8 *A, u64 want, int lane) {
    const u32 alo = (u32)(size_t)A, ahi = (u32)((size_t)A >> 32);
    const u32 lo16 = (u32)(lane < NP ? lane : NP - 1) * 16u;
#pragma unroll 1
    for (int s0 = 0; s0 < 64; s0 += BSK_PH_UNR) {
        if (((want >> s0) & ((1ULL << BSK_PH_UNR) - 1)) == 0) continue;  // wave-uniform
        u32x4 v[BSK_PH_UNR];
#pragma unroll
        for (int q = 0; q < BSK_PH_UNR; ++q) {
            const u64 as = ((u64)(u32)__builtin_amdgcn_readlane((int)ahi, s0 + q) << 32) | (u32)__builtin_amdgcn_readlane((int)alo, s0 + q);
            v[q] = *reinterpret_cast<const GLBQ u32x4_u *>(as + lo16);
        }
        if (lane < NP) {
#pragma unroll
            for (int q = 0; q < BSK_PH_UNR; ++q) *reinterpret_cast<LDSQ u32x4 *>(reg0 + (s0 + q) * (RS * 4) + 16 * lane) = v[q];
        }
    }
}

// DNA = true: the batch is 2-bit DNA and every residue dword is translated where it is fetched (DnaResidues above): no translated
// copy of the batch, no staged residue regions (the tile is the kernel's only LDS: more waves per CU).
template <int K, bool DNA = false>
__global__ __launch_bounds__(64) void k_prot_hash_fast(KArgs a) {
    typedef FastProt<1, K> FP;
    constexpr int TL = 18;
    constexpr int RS = BSK_PH_CHUNK / 4 + 4;  // dwords per staged region: 16 positions look at 8 dwords
    static_assert((RS % 4) == 0 && ((RS / 4) & 1), "region stride must be 4*odd dwords");
    constexpr int SW_OFF = 64 * TL * 8;
    __shared__ __attribute__((aligned(16))) char lds[SW_OFF + (DNA ? 64 : RS * 64 * 4)];
    LDSQ char *const lq = (LDSQ char *)lds;
    LDSQ char *const reg0 = lq + SW_OFF;
    u64 *const s_off = reinterpret_cast<u64 *>(lds);        // unit prologue only: aliases the tile
    u32 *const s_nk = reinterpret_cast<u32 *>(lds + 512);
    const int lane = lane_id();
    FP fp;  // only hashK() is used
    __shared__ u64 s_base[4];
    const int frame = a.frame;
    if (DNA) {
        DnaResidues::build_table(reinterpret_cast<u8 *>(lds + SW_OFF), a.lut, frame, lane);
        __syncthreads();
    }
    // sequence r: where it starts, its length in residues, and whether the constructor accepts it (iterator-protein.go:50: the
    // check is on the INPUT length, nucleotides for a DNA batch)
    auto span = [&](u64 r, u64 &off_, u64 &L_, u32 &Lnt_) -> bool {
        off_ = 0;
        L_ = 0;
        Lnt_ = 0;
        if (r >= a.n) return false;
        if (DNA) {
            const u64 d = a.desc[r];
            off_ = d >> 24;
            Lnt_ = (u32)(d & 0xffffffULL);
            L_ = translated_len((u64)Lnt_, frame);  // codon_tables.go:224,256
            return (u64)Lnt_ >= (u64)K * 3;
        }
        ascii_span(a, r, off_, L_);
        return prot_len_ok(a, r, L_, (u64)K * 3);
    };
    for (;;) {
      const u32 u0 = next_ticket(a.ticket, lane) * 4u;
      if (u0 >= a.nunits) break;
      const u32 u1 = u0 + 4u < a.nunits ? u0 + 4u : a.nunits;
      if (!a.uniform_len) {  // ragged batch: resolve and publish the offsets of every unit of the ticket first (see k_nthash_fast)
          u64 run = 0;
          for (u32 unit = u0; unit < u1; ++unit) {
              const u64 r = (u64)unit * 64 + lane;
              u64 L = 0, off_ = 0;
              u32 lnt_ = 0;
              const bool ok = span(r, off_, L, lnt_);
              const u32 pk = (ok && L >= (u64)K) ? ((u32)(L - K + 1) + 15u) & ~15u : 0u;
              if (lane == 0) s_base[unit - u0] = run;
              run += wave_sum_u64((u64)pk);
          }
          const u64 tbase = lookback_exclusive(a.lookback, u0 >> 2, run, lane);
          wave_sync_lds();
          if (lane < 4) s_base[lane] += tbase;
          wave_sync_lds();
      }
      for (u32 unit = u0; unit < u1; ++unit) {
        const u64 r = (u64)unit * 64 + lane;
        u64 off = 0, L = 0;
        u32 Lnt = 0;
        const bool ok = span(r, off, L, Lnt);
        const u32 nk = (ok && L >= (u64)K) ? (u32)(L - K + 1) : 0u;
        const u32 nk_max = wave_max_u32(nk);
        const u32 pk = (nk + 15u) & ~15u;
        const u64 incl = wave_incl_scan_u64((u64)pk, lane);
        const u64 T = wave_bcast_u64(incl, 63);
        const u64 base = a.uniform_len ? (u64)unit * 64 * ((nk_max + 15u) & ~15u) : s_base[unit - u0];
        const bool ovf = base + T > a.cap;
        if (ovf && lane == 0) atomicOr(&a.ticket[1], 1u);
        if (r < a.n) {
            a.refs[r] = ((base + incl - pk) << 24) | nk;
            a.status[r] = ok ? BSK_ST_OK : BSK_ST_SHORT;
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
        wave_sync_lds();
        LDSQ char *const myrow = lq + lane * (TL * 8);
        // 16 hashes of the lane through the 64 x 16 tile, out as one aligned 128-byte line per sequence
        auto hash16_and_flush = [&](const u32 (&R)[8], u32 i0) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int g = t >> 2;
                u64 h;
                switch (t & 3) {
                    case 0: h = fp.template hashK<0>(R[g], R[g + 1], R[g + 2], R[g + 3], R[g + 4]); break;
                    case 1: h = fp.template hashK<1>(R[g], R[g + 1], R[g + 2], R[g + 3], R[g + 4]); break;
                    case 2: h = fp.template hashK<2>(R[g], R[g + 1], R[g + 2], R[g + 3], R[g + 4]); break;
                    default: h = fp.template hashK<3>(R[g], R[g + 1], R[g + 2], R[g + 3], R[g + 4]); break;
                }
                *reinterpret_cast<LDSQ u64 *>(myrow + t * 8) = h;
            }
            wave_sync_lds();
            u32x4 tv[8];
#pragma unroll
            for (int rr = 0; rr < 8; ++rr)
                tv[rr] = *reinterpret_cast<LDSQ const u32x4 *>(lq + (rr * 8 + (lane >> 3)) * (TL * 8) + (lane & 7) * 16);
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
        };
        if (DNA) {
            const DnaResidues dr{a.words + off, Lnt, frame, reinterpret_cast<const LDSQ u8 *>(lq + SW_OFF)};
            u32 R[8];
            if (frame > 0) {  // wave-uniform: the aligned four-dwords-per-load form
                u32 lo4[4], hi4[4];
                dr.finish4(dr.issue4(0), lo4);
                dr.finish4(dr.issue4(4), hi4);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    R[g] = lo4[g];
                    R[g + 4] = hi4[g];
                }
                for (u32 i0 = 0; i0 < nk_max; i0 += 16) {
                    const u32x4 raw4 = dr.issue4(i0 / 4 + 8);  // the next step's four new dwords: requested before this step's hashing and stores
                    hash16_and_flush(R, i0);
                    u32 nx[4];
                    dr.finish4(raw4, nx);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        R[g] = R[g + 4];
                        R[g + 4] = nx[g];
                    }
                }
                continue;
            }
#pragma unroll
            for (int g = 0; g < 8; ++g) R[g] = dr.finish(dr.issue((u32)g), (u32)g);
            for (u32 i0 = 0; i0 < nk_max; i0 += 16) {  // minus frames: the window of four fields, aligned to the lane's own end
                const u32 jn = i0 / 4 + 8;
                bool bb;
                const u32x4 raw4 = dr.issue4m(jn, (long long)off, bb);
                const bool any_bb = __builtin_amdgcn_ballot_w64(bb) != 0;  // (only the wave of the batch's first sequence, near its start)
                u64 rawg[4] = {0, 0, 0, 0};
                if (any_bb) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) rawg[g] = dr.issue(jn + (u32)g);
                }
                hash16_and_flush(R, i0);
                u32 nx[4];
                dr.finish4m(raw4, jn, nx);
                if (any_bb) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const u32 t = dr.finish(rawg[g], jn + (u32)g);
                        nx[g] = bb ? t : nx[g];
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    R[g] = R[g + 4];
                    R[g + 4] = nx[g];
                }
            }
            continue;
        }
        const u8 *p0 = a.ascii + off;
        LDSQ char *const myreg = reg0 + lane * (RS * 4);
        for (u32 c0 = 0; c0 < nk_max; c0 += BSK_PH_CHUNK) {
            const u32 cend = (c0 + BSK_PH_CHUNK < nk_max) ? c0 + BSK_PH_CHUNK : nk_max;
            // every lane's source stays inside its own sequence (+ the buffer's slack), whatever the other lanes' lengths
            stage_chunks<RS, RS / 4>(reg0, p0 + ((u64)c0 < L ? (u64)c0 : L), __builtin_amdgcn_ballot_w64(c0 < pk), lane);
            wave_sync_lds();
            for (u32 i0 = c0; i0 < cend; i0 += 16) {
                const u32x4 ra = *reinterpret_cast<LDSQ const u32x4 *>(myreg + (i0 - c0));
                const u32x4 rb = *reinterpret_cast<LDSQ const u32x4 *>(myreg + (i0 - c0) + 16);
                const u32 R[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
                hash16_and_flush(R, i0);
            }
        }
      }
    }
}

int fast_prot_hash_short_blocks_per_cu(int k, bool dna);
void fast_prot_hash_short_launch(int k, int grid, hipStream_t stream, const KArgs &a, bool dna);
#if defined(BSK_IMPL_PROTEIN) || defined(BSK_IMPL_PROTEIN_SHORT)
#ifdef BSK_IMPL_PROTEIN_SHORT
#define BSK_PH_KS(X) X(4) X(5) X(6) X(7) X(8)
#define BSK_PH_FN(name) fast_prot_hash_short_##name
#else
#define BSK_PH_KS(X) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)
#define BSK_PH_FN(name) fast_prot_hash_long_##name
#endif
int BSK_PH_FN(blocks_per_cu)(int k, bool dna) {
    int nb = 0;
    hipError_t e = hipErrorInvalidValue;
#define X(KK)                                                                                                        \
    if (k == KK) {                                                                                                   \
        if (dna) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_prot_hash_fast<KK, true>, 64, 0);           \
        else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_prot_hash_fast<KK, false>, 64, 0);              \
    }
    BSK_PH_KS(X)
#undef X
    if (e != hipSuccess || nb < 1) {
        (void)hipGetLastError();
        nb = 1;
    }
    return nb;
}
void BSK_PH_FN(launch)(int k, int grid, hipStream_t stream, const KArgs &a, bool dna) {
#define X(KK)                                                                                                \
    if (k == KK) {                                                                                           \
        if (dna) hipLaunchKernelGGL((k_prot_hash_fast<KK, true>), dim3(grid), dim3(64), 0, stream, a);       \
        else hipLaunchKernelGGL((k_prot_hash_fast<KK, false>), dim3(grid), dim3(64), 0, stream, a);          \
    }
    BSK_PH_KS(X)
#undef X
}
#endif
#ifdef BSK_IMPL_PROTEIN
bool fast_prot_hash_supported(int k) { return k >= 4 && k <= 16; }
int fast_prot_hash_blocks_per_cu(int k) { return k <= 8 ? fast_prot_hash_short_blocks_per_cu(k, false) : fast_prot_hash_long_blocks_per_cu(k, false); }
void fast_prot_hash_launch(int k, int grid, hipStream_t stream, const KArgs &a) {
    if (k <= 8) fast_prot_hash_short_launch(k, grid, stream, a, false);
    else fast_prot_hash_long_launch(k, grid, stream, a, false);
}
int fast_prot_hash_dna_blocks_per_cu(int k) { return k <= 8 ? fast_prot_hash_short_blocks_per_cu(k, true) : fast_prot_hash_long_blocks_per_cu(k, true); }
void fast_prot_hash_dna_launch(int k, int grid, hipStream_t stream, const KArgs &a) {  // 2-bit DNA in, translated on the fly
    if (k <= 8) fast_prot_hash_short_launch(k, grid, stream, a, true);
    else fast_prot_hash_long_launch(k, grid, stream, a, true);
}
#endif  // BSK_IMPL_PROTEIN

}  // namespace bsk
