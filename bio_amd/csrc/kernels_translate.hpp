// kernels_translate.hpp -- DNA/RNA -> protein on device, as NewProteinIterator / NewProteinMinimizerSketch apply it to
// non-protein input: Translate(table, frame, trim=false, clean=false, allowUnknownCodon=true, markInitCodonAsM=false)
// (iterator-protein.go:62-67, sketch-protein.go:83-88, seq/codon_tables.go:205-285).
//
// One wavefront per unit of 64 sequences.  The unit's residue counts are a closed form of the lengths
// (codon_tables.go:224,256), their prefix comes from the look-back chain, and the unit's residues are one contiguous
// byte range of the output: sequence after sequence, the 64 lanes take 64 consecutive codons and store 64 consecutive
// bytes.  The codon tables live in LDS:
//   2-bit input : 64-entry table indexed by (first*16 + second*4 + third), codes A0 C1 G2 T3;
//   ASCII input : byte -> 4-bit IUPAC set (base2code, seq/ambiguous_bases.go:28-67; 16 = invalid letter), and the
//                 16x16x16 matrix of codonTableFromText (codon_tables.go:317-429); "---" -> '-' (codon_tables.go:167).
// Minus frames read the strand backwards and complement acgtACGT only (DNA.PairLetter, seq/alphabet.go:313-325,353-359:
// for every other letter the caller ignores the error and keeps the byte, codon_tables.go:226-228).
#pragma once
#include "device_common.hpp"

namespace bsk {

struct TArgs {
    const u32 *words;   // 2-bit input
    const u64 *desc;    // (first word << 24) | bases; NULL for a batch with a sequence of 2^24 bases or more:
    const u64 *fw, *llen;  // then first word / bases per sequence
    const u8 *ascii;    // ASCII input
    const u64 *aoff;
    u64 n;
    u32 nunits;
    int frame;          // 1,2,3,-1,-2,-3
    u64 need;           // nucleotide length below which the constructor returns ErrShortSeq (0: no flagging)
    const u8 *lut;      // device copy of the host-built tables: [0,4096) matrix, [4096,4352) base2code, [4352,4416) 2-bit table
    u8 *out;            // residues
    u64 *out_off;       // [n+1]
    u8 *short_flag;     // [n]: 1 = input shorter than `need`
    u32 *ticket;
    u64 *lookback;
    u64 *total;
};

__host__ __device__ __forceinline__ u64 translated_len(u64 L, int frame) {
    if (L < 3) return 0;  // Translate refuses (codon_tables.go:206); callers only get here with L >= 3k
    return frame > 0 ? (L - (u64)frame + 1) / 3 : (L + 1 - (u64)(-frame)) / 3;
}

template <int ENC>
__global__ __launch_bounds__(64) void k_translate(TArgs a) {
    __shared__ __attribute__((aligned(16))) u8 s_lut[4096 + 256 + 64];
    __shared__ u64 s_src[64];   // ENC 0: first word, ENC 1: first byte
    __shared__ u64 s_dst[64];
    __shared__ u32 s_len[64], s_np[64];
    const int lane = lane_id();
    {
        const u32x4 *g = reinterpret_cast<const u32x4 *>(a.lut);
        u32x4 *l = reinterpret_cast<u32x4 *>(s_lut);
        for (int i = lane; i < (4096 + 256 + 64) / 16; i += 64) l[i] = g[i];
    }
    __syncthreads();
    const int frame = a.frame;
    HeadTickets tickets(reinterpret_cast<u32 *>(a.lookback + lb_heads_at(a.nunits)));  // (one unit per ticket and a look-back over the units: device_common.hpp, "tickets from eight heads")
    for (;;) {
        const u32 unit = tickets.next(a.nunits, lane);
        if (unit == ~0u) break;
        const u64 r = (u64)unit * 64 + lane;
        u64 src = 0, L = 0;
        if (r < a.n) {
            if (ENC == 0 && a.desc) {
                const u64 d = a.desc[r];
                src = d >> 24;
                L = d & 0xffffffULL;
            } else if (ENC == 0) {
                src = a.fw[r];
                L = a.llen[r];
            } else {
                src = a.aoff[r];
                L = a.aoff[r + 1] - src;
            }
        }
        const u64 np = r < a.n ? translated_len(L, frame) : 0;
        const u64 incl = wave_incl_scan_u64(np, lane);
        const u64 T = wave_bcast_u64(incl, 63);
        const u64 base = lookback_exclusive(a.lookback, unit, T, lane);
        if (r < a.n) {
            a.out_off[r] = base + incl - np;
            a.short_flag[r] = (u8)(L < a.need ? 1 : 0);
        }
        if (unit == a.nunits - 1 && lane == 63) {
            a.out_off[a.n] = base + incl;
            *a.total = base + incl;
        }
        s_src[lane] = src;
        s_dst[lane] = base + incl - np;
        s_len[lane] = (u32)L;
        s_np[lane] = (u32)np;
        wave_sync_lds();
        for (int s = 0; s < 64; ++s) {
            const u32 P = s_np[s];
            if (P == 0) continue;  // wave-uniform
            const u64 sb = s_src[s];
            const u32 SL = s_len[s];
            u8 *dst = a.out + s_dst[s];
            for (u32 j = (u32)lane; j < P; j += 64) {
                // plus frames: codon j = bases i, i+1, i+2 with i = frame-1+3j ; minus: comp(i), comp(i-1), comp(i-2) with i = L+frame-3j
                const u32 i = frame > 0 ? (u32)(frame - 1) + 3 * j : SL - (u32)(-frame) - 3 * j;
                const u32 lo = frame > 0 ? i : i - 2;  // lowest base index of the codon
                u8 aa;
                if (ENC == 0) {
                    const u32 *w = a.words + sb + (lo >> 4);
                    const u32 six = (u32)((((u64)w[1] << 32) | w[0]) >> ((lo & 15) * 2)) & 63u;  // bases lo, lo+1, lo+2
                    const u32 b0 = six & 3, b1 = (six >> 2) & 3, b2 = six >> 4;
                    const u32 idx = frame > 0 ? (b0 << 4) | (b1 << 2) | b2 : ((b2 ^ 3) << 4) | ((b1 ^ 3) << 2) | (b0 ^ 3);
                    aa = s_lut[4096 + 256 + idx];
                } else {
                    const u8 *p = a.ascii + sb + lo;
                    u8 c0 = p[0], c1 = p[1], c2 = p[2];
                    if (frame < 0) {
                        const u8 t = c0;
                        c0 = c2;
                        c2 = t;
                        auto pair = [](u8 b) -> u8 {
                            switch (b) {
                                case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
                                case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
                                default: return b;
                            }
                        };
                        c0 = pair(c0);
                        c1 = pair(c1);
                        c2 = pair(c2);
                    }
                    const u32 x = s_lut[4096 + c0], y = s_lut[4096 + c1], z = s_lut[4096 + c2];
                    if ((x | y | z) & 16u) aa = 'X';                              // a letter outside base2code (codon_tables.go:160-163)
                    else if (c0 == '-' && c1 == '-' && c2 == '-') aa = '-';     // codon_tables.go:167
                    else aa = s_lut[(x << 8) | (y << 4) | z];                     // 0 was replaced by 'X' on the host (:172-174)
                }
                dst[j] = aa;
            }
        }
        wave_sync_lds();
    }
}

}  // namespace bsk
