// kernels_more.hpp -- the remaining iterator kinds of sketches/ on the general mapping
// (one read per lane, rolling per lane, LDS-staged CSR output):
//   BSK_SYNCMER         NewSyncmerSketch / NextSyncmer           sketch.go:142-202, 312-477
//   BSK_PROT_HASH       NewProteinIterator / Next                iterator-protein.go:46-90
//   BSK_PROT_MINIMIZER  NewProteinMinimizerSketch / Next         sketch-protein.go:62-210
//   BSK_KMER            NewKmerIterator / NextKmer               iterator.go:668-759
//   BSK_SIMHASH         NewSimHashIterator / NextSimHash         iterator.go:113-612
// Correctness-first kernels: bit-exact, any parameters; the specialised fast paths are
// in kernels_fast.hpp.
#pragma once
#include "kernels_generic.hpp"

namespace bsk {

// ---------------------------------------------------------------------------------------
// wyhash "version 1" as ported by github.com/zeebo/wyhash v0.0.1 (go.mod:15; un-vendored:
// PARITY UNPINNED).  Called with seed 1 on every k residues
// (iterator-protein.go:87, sketch-protein.go:117).  Bytes are assembled one by one, so any
// alignment / any k works.
// ---------------------------------------------------------------------------------------
#define WYP0 0xa0761d6478bd642fULL
#define WYP1 0xe7037ed1a0b428dbULL
#define WYP2 0x8ebc6af09c88c6e3ULL
#define WYP3 0x589965cc75374cc3ULL
#define WYP4 0x1d8e4e27c47d124fULL
#define WYP5 0xeb44accab455d165ULL
__device__ __forceinline__ u64 wymum(u64 a, u64 b) { return __umul64hi(a, b) ^ (a * b); }
__device__ __forceinline__ u64 wyr08(const u8 *p) { return p[0]; }
__device__ __forceinline__ u64 wyr16(const u8 *p) { return (u64)p[0] | ((u64)p[1] << 8); }
__device__ __forceinline__ u64 wyr32(const u8 *p) { return wyr16(p) | (wyr16(p + 2) << 16); }
__device__ __forceinline__ u64 wyr64(const u8 *p) { return wyr32(p) | (wyr32(p + 4) << 32); }
__device__ __forceinline__ u64 wyr64s(const u8 *p) { return (wyr32(p) << 32) | wyr32(p + 4); }
// tail of 1..7 bytes packed big-end-first exactly as the reference switch does
__device__ __forceinline__ u64 wytail(const u8 *p, unsigned n) {
    switch (n) {
        case 1: return wyr08(p);
        case 2: return wyr16(p);
        case 3: return (wyr16(p) << 8) | wyr08(p + 2);
        case 4: return wyr32(p);
        case 5: return (wyr32(p) << 8) | wyr08(p + 4);
        case 6: return (wyr32(p) << 16) | wyr16(p + 4);
        case 7: return (wyr32(p) << 24) | (wyr16(p + 4) << 8) | wyr08(p + 6);
        default: return wyr64s(p);  // 8
    }
}
__device__ inline u64 wyhash_dev(const u8 *key, unsigned len, u64 seed) {
    const u8 *p = key;
    unsigned i;
    for (i = 0; i + 32 <= len; i += 32, p += 32)
        seed = wymum(seed ^ WYP0, wymum(wyr64(p) ^ WYP1, wyr64(p + 8) ^ WYP2) ^ wymum(wyr64(p + 16) ^ WYP3, wyr64(p + 24) ^ WYP4));
    seed ^= WYP0;
    const unsigned r = len & 31;
    if (r >= 1 && r <= 8) seed = wymum(seed, wytail(p, r) ^ WYP1);
    else if (r <= 16 && r) seed = wymum(wyr64s(p) ^ seed, wytail(p + 8, r - 8) ^ WYP2);
    else if (r <= 24 && r) seed = wymum(wyr64s(p) ^ seed, wyr64s(p + 8) ^ WYP2) ^ wymum(seed, wytail(p + 16, r - 16) ^ WYP3);
    else if (r) seed = wymum(wyr64s(p) ^ seed, wyr64s(p + 8) ^ WYP2) ^ wymum(wyr64s(p + 16) ^ seed, wytail(p + 24, r - 24) ^ WYP4);
    return wymum(seed, (u64)len ^ WYP5);
}

// Length rule of the protein constructors (iterator-protein.go:50, sketch-protein.go:66,73): checked on the INPUT.
// Protein input: the residues themselves.  Translated input: the translate kernel left one flag per sequence
// (rflags: 1 = too few nucleotides); the translation itself may then hold fewer than k residues (or fewer than w
// k-mers), which yields nothing but is no error.
__device__ __forceinline__ bool prot_len_ok(const KArgs &a, u64 r, u64 L, u64 need) {
    return a.rflags ? a.rflags[r] == 0 : L >= need;
}

struct WySrc {  // hash source over residues
    const u8 *a;
    u64 L;
    int k;
    __device__ __forceinline__ void init(const u8 *ascii, u64 off, u64 len, int k_) {
        a = ascii + off;
        L = len;
        k = k_;
    }
    __device__ __forceinline__ void step(u32 i, u64 &h, u32 &rev) {
        rev = 0;
        h = ((u64)i + (u64)k <= L) ? wyhash_dev(a + i, (unsigned)k, 1) : 0;
    }
};

// ---------------------------------------------------------------------------------------
// PROT_MINIMIZER: the window machine of kernels_generic.hpp fed by WySrc.
// ---------------------------------------------------------------------------------------
static __global__ __launch_bounds__(64) void k_prot_minimizer(KArgs a) {
    constexpr int CAP = BSK_GEN_CAP;
    __shared__ u64 s_h[CAP * 64];
    __shared__ u32 s_p[CAP * 64];
    __shared__ u16 s_m[CAP * 64];
    const int lane = lane_id();
    Stage<CAP> st{s_h, s_p, s_m};
    u64 *ring_h = a.ring_h + (u64)blockIdx.x * a.ring_w * 64;
    u32 *ring_p = a.ring_p + (u64)blockIdx.x * a.ring_w * 64;
    const int W = a.w;
    for (;;) {
        const u32 unit = next_ticket(a.ticket, lane);
        if (unit >= a.nunits) break;
        const u64 r = (u64)unit * 64 + lane;
        u64 off = 0, L = 0;
        if (r < a.n) {
            ascii_span(a, r, off, L);
        }
        // sketch-protein.go:66,73: len < 3k -> ErrShortSeq ; len < 3k+w-1 -> ErrShortSeq (on the INPUT length)
        const bool ok = r < a.n && prot_len_ok(a, r, L, (u64)a.k * 3 + (u64)W - 1);
        const u32 nk = (ok && L >= (u64)a.k + (u64)W - 1) ? (u32)(L - a.k + 1) : 0u;
        const u32 nk_max = wave_max_u32(nk);
        u32 cnt = 0, tie = 0;
        WySrc src;
        if (nk_max) {
            src.init(a.ascii, off, L, a.k);
            window_pass<WySrc, CAP, false>(src, nk, nk_max, W, ring_h, ring_p, lane, st, cnt, tie, a.hash, a.pos, 0);
        }
        u32 excl;
        bool ovf;
        const u64 base = unit_epilogue<CAP>(a, unit, lane, r, cnt, st, excl, ovf);
        if (!ovf && __ballot(cnt > (u32)CAP)) {
            u32 c2 = 0, t2 = 0;
            src.init(a.ascii, off, L, a.k);
            window_pass<WySrc, CAP, true>(src, nk, nk_max, W, ring_h, ring_p, lane, st, c2, t2, a.hash, a.pos, base + excl);
        }
        if (r < a.n) a.status[r] = (u8)((ok ? BSK_ST_OK : BSK_ST_SHORT) | ((tie && ok) ? BSK_ST_FIRST_WINDOW_TIE : 0));
    }
}

// ---------------------------------------------------------------------------------------
// Generic "every position" stream: value i of read r -> hash[first(r) + i], through the
// 64x16 LDS transpose tile.  Src: init() done by the caller, step(i, h, rev).
// ---------------------------------------------------------------------------------------
template <class Src>
__device__ __forceinline__ void stream_values(Src &src, const KArgs &a, u32 nk_max, int lane, u64 *s_tile, const u64 *s_off,
                                              const u32 *s_nk) {
    for (u32 i = 0; i < nk_max; ++i) {
        u64 h;
        u32 rev;
        src.step(i, h, rev);
        s_tile[lane * TILE_LD + (i & 15)] = h;
        if ((i & 15) == 15 || i == nk_max - 1) {
            wave_sync_lds();
            const u32 c0 = i & ~15u;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int row = rr * 8 + (lane >> 3);
                const u32 col = (u32)(lane & 7) * 2;
                const u32 ia = c0 + col;
                const u32 nkr = s_nk[row];
                u64 *dst = a.hash + s_off[row] + ia;
                if (ia < nkr) dst[0] = s_tile[row * TILE_LD + col];
                if (ia + 1 < nkr) dst[1] = s_tile[row * TILE_LD + col + 1];
            }
            wave_sync_lds();
        }
    }
}

// per-unit bookkeeping shared by the stream kernels: counts -> look-back -> refs/status; returns false if nothing to do
__device__ __forceinline__ bool stream_prologue(const KArgs &a, u32 unit, int lane, u64 r, u32 nk, u8 sbyte, u64 *s_off, u32 *s_nk,
                                                u32 &nk_max) {
    nk_max = wave_max_u32(nk);
    const u64 incl = wave_incl_scan_u64((u64)nk, lane);
    const u64 T = wave_bcast_u64(incl, 63);
    u64 first = 0;
    bool ovf = false;
    if (a.inplace) {  // side launch: the run the main launch reserved for this read
        if (r < a.n) first = a.refs[r] >> 24;
    } else {
        const u64 base = a.out_base + lookback_exclusive(a.lookback, unit, T, lane);
        ovf = base + T > a.cap;
        if (ovf && lane == 0) atomicOr(&a.ticket[1], 1u);
        if (unit == a.nunits - 1 && lane == 63) *a.total = base + incl;
        first = base + incl - nk;
    }
    if (r < a.n) {
        a.refs[r] = (first << 24) | nk;
        a.status[r] = sbyte;
    }
    if (ovf || nk_max == 0) return false;
    s_off[lane] = first;
    s_nk[lane] = nk;
    wave_sync_lds();
    return true;
}

// ---- PROT_HASH ----
static __global__ __launch_bounds__(64) void k_prot_hash(KArgs a) {
    __shared__ u64 s_tile[64 * TILE_LD];
    __shared__ u64 s_off[64];
    __shared__ u32 s_nk[64];
    const int lane = lane_id();
    for (;;) {
        const u32 unit = next_ticket(a.ticket, lane);
        if (unit >= a.nunits) break;
        const u64 r = (u64)unit * 64 + lane;
        u64 off = 0, L = 0;
        if (r < a.n) {
            ascii_span(a, r, off, L);
        }
        const bool ok = r < a.n && prot_len_ok(a, r, L, (u64)a.k * 3);  // iterator-protein.go:50 (checked on the input length)
        const u32 nk = (ok && L >= (u64)a.k) ? (u32)(L - a.k + 1) : 0u;
        u32 nk_max;
        if (!stream_prologue(a, unit, lane, r, nk, ok ? BSK_ST_OK : BSK_ST_SHORT, s_off, s_nk, nk_max)) continue;
        WySrc src;
        src.init(a.ascii, off, L, a.k);
        stream_values(src, a, nk_max, lane, s_tile, s_off, s_nk);
    }
}

// ---------------------------------------------------------------------------------------
// KMER: 2-bit codes, first base in the most significant pair (iterator.go:736,740).
// base2bit = sketches/kmers.go:23-40 (IUPAC letters map to one member, everything else 4).
// canonical: min(code, rc).  Non-canonical: the forward-strand codes, then the codes of the
// reverse-complemented sequence (iterator.go:713-723), i.e. rc codes in reverse position order.
// An illegal base ends the read's stream at the first k-mer that contains it (status ILLEGAL).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned base2bit_dev(unsigned b) {
    switch (b | 0x20) {  // table is case-insensitive
        case 'a': case 'd': case 'h': case 'm': case 'n': case 'r': case 'v': case 'w': return 0;
        case 'b': case 'c': case 's': case 'y': return 1;
        case 'g': case 'k': return 2;
        case 't': case 'u': return 3;
        default: return 4;
    }
}

// complement letter (Alphabet.PairLetter, seq/alphabet.go:313-325) of the batch's alphabet `pairs` (bsk_alphabet): DNAredundant (0),
// DNA (2), RNA (3), RNAredundant (4) -- seq/alphabet.go:353-383 -- or Unlimit (5: ComplementInplace returns at once,
// seq/seq.go:381-383).  Letters without a pair stay (the error is dropped, seq/seq.go:390).
__device__ __forceinline__ unsigned dna_pair_dev(unsigned b, int pairs) {
    if (pairs == BSK_ALPHA_UNLIMIT) return b;
    const unsigned lower = b & 0x20, u = b & ~0x20u;
    const bool rna = pairs == BSK_ALPHA_RNA || pairs == BSK_ALPHA_RNA_REDUNDANT;
    const bool redundant = pairs == BSK_ALPHA_DNA || pairs == BSK_ALPHA_RNA_REDUNDANT;
    unsigned p;
    switch (u) {
        case 'A': p = rna ? 'U' : 'T'; break;
        case 'C': p = 'G'; break;
        case 'G': p = 'C'; break;
        case 'T': if (rna) return b; p = 'A'; break;
        case 'U': if (!rna) return b; p = 'A'; break;
        default:
            if (!redundant) return b;
            p = 0;
    }
    if (p) return p | lower;
    switch (u) {
        case 'R': p = 'Y'; break;
        case 'Y': p = 'R'; break;
        case 'K': p = 'M'; break;
        case 'M': p = 'K'; break;
        case 'B': p = 'V'; break;
        case 'V': p = 'B'; break;
        case 'D': p = 'H'; break;
        case 'H': p = 'D'; break;
        default: return b;  // S, W, N, gaps, anything else: unchanged
    }
    return p | lower;
}

// the same on 2-bit codes (A0 C1 G2 T3) of a pure-ACGT sequence, as four 2-bit entries: DNA 3,2,1,0; RNA 3,2,1,3 (a 'T' has no pair and
// stays); Unlimit 0,1,2,3
__host__ __device__ __forceinline__ unsigned pair_map2(int pairs) {
    if (pairs == BSK_ALPHA_UNLIMIT) return 0xE4u;
    if (pairs == BSK_ALPHA_RNA || pairs == BSK_ALPHA_RNA_REDUNDANT) return 0xDBu;
    return 0x1Bu;
}

template <int ENC>
struct KmerSrc {
    const u32 *w;
    const u8 *a;
    u64 L;
    int k, canonical, pairs;
    unsigned map2;
    u64 code, rc, rc2, mask1;  // rc: arithmetic complement (canonical, iterator.go:740); rc2: code of the
                               // reverse-COMPLEMENTED LETTERS (second strand of the non-canonical mode, iterator.go:719)
    unsigned sh2;
    __device__ __forceinline__ unsigned base(u64 t) const {
        if (ENC) return t < L ? base2bit_dev(a[t]) & 3u : 0u;
        return (w[t >> 4] >> ((t & 15) * 2)) & 3u;
    }
    __device__ __forceinline__ unsigned cbase(u64 t, unsigned b) const {  // 2-bit code of the complement letter
        if (ENC) return t < L ? base2bit_dev(dna_pair_dev(a[t], pairs)) & 3u : 0u;
        return (map2 >> (2u * b)) & 3u;
    }
    __device__ __forceinline__ void init(const u32 *words, const u8 *ascii, u64 off, u64 len, int k_, int canon, int pairs_ = 0) {
        pairs = pairs_;
        map2 = pair_map2(pairs_);
        w = words + (ENC ? 0 : off);
        a = ascii + (ENC ? off : 0);
        L = len;
        k = k_;
        canonical = canon;
        mask1 = (k_ >= 33) ? ~0ULL : ((1ULL << (2 * (k_ - 1))) - 1ULL);
        sh2 = 2u * (unsigned)(k_ - 1);
        code = rc = rc2 = 0;
        for (int j = 0; j < k_ - 1; ++j) {  // first k-1 bases (kmers.Encode + MustRevComp, iterator.go:742-743)
            const u64 b = base((u64)j);
            code = (code << 2) | b;
            rc = (rc >> 2) | ((b ^ 3) << sh2);
            rc2 = (rc2 >> 2) | ((u64)cbase((u64)j, (unsigned)b) << sh2);
        }
    }
    __device__ __forceinline__ void step2(u32 i, u64 &fwd, u64 &rev_) {
        const u64 b = base((u64)i + (u64)k - 1);
        code = ((code & mask1) << 2) | b;   // iterator.go:736
        rc = ((b ^ 3) << sh2) | (rc >> 2);  // iterator.go:740
        rc2 = ((u64)cbase((u64)i + (u64)k - 1, (unsigned)b) << sh2) | (rc2 >> 2);
        fwd = code;
        rev_ = rc;
    }
    __device__ __forceinline__ void step(u32 i, u64 &h, u32 &rev) {
        u64 f, r2;
        step2(i, f, r2);
        rev = (canonical && r2 < f) ? 1u : 0u;  // iterator.go:754
        h = rev ? r2 : f;
    }
};

template <int ENC>
__global__ __launch_bounds__(64) void k_kmer(KArgs a) {
    __shared__ u64 s_tile[64 * TILE_LD];
    __shared__ u64 s_tile2[64 * TILE_LD];
    __shared__ u64 s_off[64];
    __shared__ u32 s_nk[64];
    const int lane = lane_id();
    for (;;) {
        const u32 unit = next_ticket(a.ticket, lane);
        if (unit >= a.nunits) break;
        const u64 r = read_index(a, unit, lane);
        u64 off = 0, L = 0;
        if (r < a.n) {
            if (ENC) {
                ascii_span(a, r, off, L);
            } else {
                const u64 d = a.desc[r];
                off = d >> 24;
                L = d & 0xffffffULL;
            }
        }
        const bool ok = r < a.n && L >= (u64)a.circ_ext && (L - (u64)a.circ_ext) >= (u64)a.k;  // iterator.go:672
        u32 nk = ok ? (u32)(L - a.k + 1) : 0u;
        u8 sbyte = ok ? BSK_ST_OK : BSK_ST_SHORT;
        if (ENC && ok) {  // first illegal base b ends the stream before k-mer max(0, b-k+1) (iterator.go:731-733,746-748)
            u64 bad = L;
            for (u64 t = 0; t < L; ++t)
                if (base2bit_dev(a.ascii[off + t]) > 3) {
                    bad = t;
                    break;
                }
            if (bad < L) {
                sbyte = BSK_ST_ILLEGAL;
                const u64 lim = bad + 1 >= (u64)a.k ? bad + 1 - (u64)a.k : 0;
                if (lim < nk) nk = (u32)lim;
            }
        }
        if (ok && a.rflags && !ENC) sbyte |= a.rflags[r];
        const bool two = !a.canonical && !a.one_strand && sbyte != BSK_ST_ILLEGAL;  // the reverse strand is only reached without an error
        const u32 nvals = two ? 2u * nk : nk;
        u32 nv_max;
        if (!stream_prologue(a, unit, lane, r, nvals, sbyte, s_off, s_nk, nv_max)) continue;
        const u32 nk_max = wave_max_u32(nk);
        KmerSrc<ENC> src;
        src.init(a.words, a.ascii, off, L, a.k, a.canonical, a.pairs);
        const bool any_two = __ballot(two) != 0;
        for (u32 i = 0; i < nk_max; ++i) {
            u64 f, rcv;
            src.step2(i, f, rcv);
            u64 h = f;
            if (a.canonical && rcv < f) h = rcv;
            s_tile[lane * TILE_LD + (i & 15)] = h;
            s_tile2[lane * TILE_LD + 15 - (i & 15)] = src.rc2;  // reverse strand is emitted in reverse position order
            if ((i & 15) == 15 || i == nk_max - 1) {
                wave_sync_lds();
                const u32 c0 = i & ~15u;
                for (int rr = 0; rr < 8; ++rr) {
                    const int row = rr * 8 + (lane >> 3);
                    const u32 nkr = s_nk[row];  // values of that row: nk or 2*nk
                    for (int c = 0; c < 2; ++c) {
                        const u32 col = (u32)(lane & 7) * 2 + c;
                        const u32 ia = c0 + col;
                        // forward value i -> index i (valid while i < that row's k-mer count)
                        const u32 row_two = __shfl((u32)two, row, 64);
                        const u32 nkk = row_two ? nkr / 2 : nkr;
                        if (ia < nkk) a.hash[s_off[row] + ia] = s_tile[row * TILE_LD + col];
                        if (any_two && row_two) {
                            // tile2 column c' holds k-mer i = c0 + 15 - c' ; its output index is 2*nkk - 1 - i
                            const u32 ik = c0 + 15 - col;
                            if (ik < nkk) a.hash[s_off[row] + (2 * nkk - 1 - ik)] = s_tile2[row * TILE_LD + col];
                        }
                    }
                }
                wave_sync_lds();
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// SIMHASH (iterator.go:191-612): per k-mer, the k-m+1 canonical/forward ntHashes of its m-mers,
// FracMinHash-filtered (hash > MaxUint64/scale -> absent), bit b of the code = 1 iff at least
// (nPos+1)/2 of the present hashes have bit b set.  Ring of the m-mer hashes in global scratch
// ([slot][lane]); the 64 per-bit counters in registers.
// ---------------------------------------------------------------------------------------
template <int ENC>
__global__ __launch_bounds__(64) void k_simhash(KArgs a) {
    __shared__ uint4 s_tab[ENC ? 512 : 32];
    __shared__ u64 s_tile[64 * TILE_LD];
    __shared__ u64 s_off[64];
    __shared__ u32 s_nk[64];
    const int lane = lane_id();
    if (ENC) build_bytetabs(s_tab, s_tab + 256, a.m, lane);
    else build_xtab(s_tab, a.m, lane);
    __syncthreads();
    u64 *ring = a.ring_h + (u64)blockIdx.x * a.ring_w * 64;
    const int nh = a.k - a.m + 1;
    const u64 maxhash = a.scale > 1 ? 0xffffffffffffffffULL / (u64)a.scale : 0xffffffffffffffffULL;
    for (;;) {
        const u32 unit = next_ticket(a.ticket, lane);
        if (unit >= a.nunits) break;
        const u64 r = read_index(a, unit, lane);
        u64 off = 0, L = 0;
        if (r < a.n) {
            if (ENC) {
                ascii_span(a, r, off, L);
            } else {
                const u64 d = a.desc[r];
                off = d >> 24;
                L = d & 0xffffffULL;
            }
        }
        const bool ok = r < a.n && L >= (u64)a.circ_ext && (L - (u64)a.circ_ext) >= (u64)a.k;  // iterator.go:128
        const u32 nk = ok ? (u32)(L - a.k + 1) : 0u;
        u8 sbyte = ok ? BSK_ST_OK : BSK_ST_SHORT;
        if (ok && a.rflags) sbyte |= a.rflags[r];
        u32 nk_max;
        if (!stream_prologue(a, unit, lane, r, nk, sbyte, s_off, s_nk, nk_max)) continue;
        NtPacked sp;
        NtAscii sa;
        if (ENC) sa.init(a.ascii, off, L, a.m, a.canonical, s_tab, s_tab + 256);
        else sp.init(a.words, off, a.m, a.canonical, s_tab);
        int sum[64];
#pragma unroll
        for (int b = 0; b < 64; ++b) sum[b] = 0;
        int npos = 0;
        u32 mi = 0;  // next m-mer index to hash
        auto next_m = [&]() -> u64 {
            u64 h;
            u32 rv;
            if (ENC) sa.step(mi, h, rv);
            else sp.step(mi, h, rv);
            ++mi;
            return h > maxhash ? 0ULL : h;  // iterator.go:281-285,445-448
        };
        auto add = [&](u64 h, int d) {
#pragma unroll
            for (int b = 0; b < 64; ++b) sum[b] += d * (int)((h >> (63 - b)) & 1);
        };
        int pre_i = 0;
        for (u32 i = 0; i < nk_max; ++i) {
            if (i == 0) {
                for (int j = 0; j < nh; ++j) {  // iterator.go:441-523
                    const u64 h = next_m();
                    ring[j * 64 + lane] = h;
                    if (h) {
                        npos++;
                        add(h, 1);
                    }
                }
                pre_i = 0;
            } else {
                const u64 old = ring[pre_i * 64 + lane];  // iterator.go:204
                if (old) {
                    npos--;
                    add(old, -1);
                }
                const u64 h = next_m();
                ring[pre_i * 64 + lane] = h;
                if (h) {
                    npos++;
                    add(h, 1);
                }
                pre_i = (pre_i == nh - 1) ? 0 : pre_i + 1;  // iterator.go:434-438
            }
            u64 code = 0;
            const int thr = (npos + 1) / 2;  // iterator.go:360
            if (npos > 0) {
#pragma unroll
                for (int b = 0; b < 64; ++b) code |= (u64)(sum[b] >= thr ? 1 : 0) << (63 - b);
            }
            s_tile[lane * TILE_LD + (i & 15)] = code;
            if ((i & 15) == 15 || i == nk_max - 1) {
                wave_sync_lds();
                const u32 c0 = i & ~15u;
                for (int rr = 0; rr < 8; ++rr) {
                    const int row = rr * 8 + (lane >> 3);
                    const u32 col = (u32)(lane & 7) * 2;
                    const u32 ia = c0 + col;
                    const u32 nkr = s_nk[row];
                    u64 *dst = a.hash + s_off[row] + ia;
                    if (ia < nkr) dst[0] = s_tile[row * TILE_LD + col];
                    if (ia + 1 < nkr) dst[1] = s_tile[row * TILE_LD + col + 1];
                }
                wave_sync_lds();
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// SYNCMER (the reference's window-bounded closed syncmer, sketch.go:312-477; closed form in
// DESIGN.md): w = k-s; for idx in [0, end], end = L-2k+s+1: mI = leftmost argmin of the canonical
// s-mer hashes at positions [idx, idx+2w-1]; b = mI if mI-idx < w else mI-w; every distinct b <= end
// is emitted once, in increasing order, as the canonical k-mer hash at b.
// Pass 1 (s-mers, window 2w) stages the selected POSITIONS; pass 2 (k-mers) fills in the hashes.
// ---------------------------------------------------------------------------------------
template <class Src, int CAP, bool DIRECT>
__device__ __forceinline__ void syncmer_pass1(Src &src, u32 ns, u32 ns_max, int W2, int w, long long end, u64 *ring_h, u32 *ring_p,
                                              int lane, Stage<CAP> st, u32 &cnt, u32 &tie, u32 *gpos, u64 gbase) {
    u64 Ph = 0;
    u32 Pp = 0, prevb = 0xffffffffu;
    int o = 0;
    bool first = true;
    for (u32 i = 0; i < ns_max; ++i) {
        u64 h;
        u32 rev;
        src.step(i, h, rev);
        const bool act = i < ns;
        if (o == 0 || h < Ph) {
            Ph = h;
            Pp = i;
        }
        if (!first || o == W2 - 1) {
            u32 mp = Pp;
            if (o != W2 - 1) {
                const u64 Sh = ring_h[(o + 1) * 64 + lane];
                if (!(Ph < Sh)) mp = ring_p[(o + 1) * 64 + lane];
            }
            const u32 idx = i + 1 - (u32)W2;                                // window start
            const u32 b = (mp - idx < (u32)w) ? mp : mp - (u32)w;           // sketch.go:413-420
            const bool emit = act && b != prevb && (long long)b <= end;     // selections beyond `end` are never reached (sketch.go:314)
            if (act) prevb = b;
            if (emit) {
                if (!DIRECT) {
                    if (cnt < (u32)CAP) st.sp[Stage<CAP>::slot(cnt, lane)] = b;
                } else {
                    gpos[gbase + cnt] = b;
                }
                cnt++;
            }
        }
        ring_h[o * 64 + lane] = h;
        ring_p[o * 64 + lane] = i;
        if (o == W2 - 1) {
            u64 nh = h;
            u32 np = i;
            u32 dup = 0;  // BSK_ST_FIRST_WINDOW_TIE over the first 2w s-mers (kernels_fast.hpp, suffix_min_pass)
            for (int q = W2 - 2; q >= 0; --q) {
                const u64 ah = ring_h[q * 64 + lane];
                const u32 ap = ring_p[q * 64 + lane];
                if (nh < ah) {
                    ring_h[q * 64 + lane] = nh;
                    ring_p[q * 64 + lane] = np;
                } else {
                    dup = ah == nh ? 1u : 0u;
                    nh = ah;
                    np = ap;
                }
                if (first && !DIRECT) tie |= dup;
            }
            o = 0;
            first = false;
        } else {
            ++o;
        }
    }
}

template <int ENC>
__global__ __launch_bounds__(64) void k_syncmer(KArgs a) {
    constexpr int CAP = BSK_GEN_CAP;
    __shared__ uint4 s_tabk[ENC ? 512 : 32];
    __shared__ uint4 s_tabs[ENC ? 512 : 32];
    __shared__ u64 s_h[CAP * 64];
    __shared__ u32 s_p[CAP * 64];
    __shared__ u16 s_m[CAP * 64];
    const int lane = lane_id();
    if (ENC) {
        build_bytetabs(s_tabk, s_tabk + 256, a.k, lane);
        build_bytetabs(s_tabs, s_tabs + 256, a.s, lane);
    } else {
        build_xtab(s_tabk, a.k, lane);
        build_xtab(s_tabs, a.s, lane);
    }
    __syncthreads();
    Stage<CAP> st{s_h, s_p, s_m};
    u64 *ring_h = a.ring_h + (u64)blockIdx.x * a.ring_w * 64;
    u32 *ring_p = a.ring_p + (u64)blockIdx.x * a.ring_w * 64;
    const int w = a.k - a.s, W2 = 2 * w;
    const bool skip = a.s == a.k;  // sketch.go:328: every k-mer
    for (;;) {
        const u32 unit = next_ticket(a.ticket, lane);
        if (unit >= a.nunits) break;
        const u64 r = read_index(a, unit, lane);
        u64 off = 0, L = 0;
        if (r < a.n) {
            if (ENC) {
                ascii_span(a, r, off, L);
            } else {
                const u64 d = a.desc[r];
                off = d >> 24;
                L = d & 0xffffffULL;
            }
        }
        // sketch.go:149: len(S.Seq) < 2k-s-1 -> ErrShortSeq (un-extended length); a read shorter than k cannot be hashed at all
        const long long Lorig = (long long)L - a.circ_ext;
        const bool ok = r < a.n && Lorig >= 0 && Lorig >= 2LL * a.k - a.s - 1 && L >= (u64)a.k;
        const long long end = (long long)L - 2LL * a.k + a.s + 1;  // sketch.go:170
        const u32 nkk = ok ? (u32)(L - a.k + 1) : 0u;              // k-mers
        const u32 ns = (ok && !skip) ? (u32)(L - a.s + 1) : 0u;    // s-mers
        const u32 ns_max = wave_max_u32(ns), nkk_max = wave_max_u32(nkk);
        u32 cnt = 0, tie = 0;
        NtPacked sp;
        NtAscii sa;
        if (skip) {
            cnt = nkk;  // every k-mer is selected: stage positions 0..cnt-1 as pass 1 would (when they fit)
            if (!__ballot(cnt > (u32)CAP)) {
                for (u32 e = 0; e < nkk_max; ++e)
                    if (e < cnt) st.sp[Stage<CAP>::slot(e, lane)] = e;
            }
        } else if (ns_max) {
            if (ENC) {
                sa.init(a.ascii, off, L, a.s, 1, s_tabs, s_tabs + 256);
                syncmer_pass1<NtAscii, CAP, false>(sa, ns, ns_max, W2, w, end, ring_h, ring_p, lane, st, cnt, tie, a.pos, 0);
            } else {
                sp.init(a.words, off, a.s, 1, s_tabs);
                syncmer_pass1<NtPacked, CAP, false>(sp, ns, ns_max, W2, w, end, ring_h, ring_p, lane, st, cnt, tie, a.pos, 0);
            }
        }
        const bool direct = __ballot(cnt > (u32)CAP) != 0;  // wave-uniform
        // pass 2 (staged): walk the k-mers, fill in hash + strand at the selected positions
        if (!direct && nkk_max) {
            if (ENC) sa.init(a.ascii, off, L, a.k, 1, s_tabk, s_tabk + 256);
            else sp.init(a.words, off, a.k, 1, s_tabk);
            u32 e = 0;
            u32 nextp = cnt ? st.sp[Stage<CAP>::slot(0, lane)] : 0xffffffffu;
            u32 last = cnt ? st.sp[Stage<CAP>::slot(cnt - 1, lane)] + 1 : 0;
            const u32 steps = wave_max_u32(last);
            for (u32 i = 0; i < steps; ++i) {
                u64 h;
                u32 rev;
                if (ENC) sa.step(i, h, rev);
                else sp.step(i, h, rev);
                if (i == nextp) {
                    const u32 sl = Stage<CAP>::slot(e, lane);
                    st.sh[sl] = h;
                    st.sp[sl] = i | (rev << 31);
                    ++e;
                    nextp = e < cnt ? st.sp[Stage<CAP>::slot(e, lane)] : 0xffffffffu;
                }
            }
        }
        u32 excl;
        bool ovf;
        const u64 base = unit_epilogue<CAP>(a, unit, lane, r, cnt, st, excl, ovf);
        if (!ovf && direct && nkk_max) {
            // rare / skip mode: positions (pass 1) and hashes (pass 2) go straight to HBM
            const u64 gb = base + excl;
            if (!skip) {
                u32 c2 = 0, t2 = 0;
                if (ENC) {
                    sa.init(a.ascii, off, L, a.s, 1, s_tabs, s_tabs + 256);
                    syncmer_pass1<NtAscii, CAP, true>(sa, ns, ns_max, W2, w, end, ring_h, ring_p, lane, st, c2, t2, a.pos, gb);
                } else {
                    sp.init(a.words, off, a.s, 1, s_tabs);
                    syncmer_pass1<NtPacked, CAP, true>(sp, ns, ns_max, W2, w, end, ring_h, ring_p, lane, st, c2, t2, a.pos, gb);
                }
            }
            if (ENC) sa.init(a.ascii, off, L, a.k, 1, s_tabk, s_tabk + 256);
            else sp.init(a.words, off, a.k, 1, s_tabk);
            u32 e = 0;
            for (u32 i = 0; i < nkk_max; ++i) {
                u64 h;
                u32 rev;
                if (ENC) sa.step(i, h, rev);
                else sp.step(i, h, rev);
                if (e < cnt) {
                    const u32 want = skip ? i : (a.pos[gb + e] & BSK_POS_MASK);
                    if (i == want && i < nkk) {
                        a.hash[gb + e] = h;
                        a.pos[gb + e] = i | (rev << 31);
                        ++e;
                    }
                }
            }
        }
        if (r < a.n) {
            u8 sbyte = ok ? BSK_ST_OK : BSK_ST_SHORT;
            if (tie && ok) sbyte |= BSK_ST_FIRST_WINDOW_TIE;
            if (ok && a.rflags) sbyte |= a.rflags[r];
            a.status[r] = sbyte;
        }
    }
}

}  // namespace bsk
