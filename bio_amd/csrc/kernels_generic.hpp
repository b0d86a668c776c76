// kernels_generic.hpp -- the general ("any k, any w, any byte") biosketch kernels.
//
// Mapping: ONE READ PER LANE.  A wavefront owns a *unit* of 64 consecutive reads
// and walks all of them in lock-step, one k-mer per step, with a true rolling
// ntHash per lane (the reference's own recurrence, iterator.go:658-665 ->
// nthash.Next).  All loop counters are wave-uniform (SGPRs); only data differs
// per lane.  Selected tuples are staged per lane in LDS and written out as one
// contiguous, position-ordered block per unit (CSR), whose global offset comes
// from a decoupled look-back over the unit totals.
//
// These kernels are the correctness baseline on the device and the fall-back for
// parameters the specialised kernels (kernels_fast.hpp) do not cover.
#pragma once
#include "device_common.hpp"
#include "biosketch.h"

namespace bsk {

struct KArgs {
    // input batch
    const u32 *words;   // packed DNA (16 bases / word)
    const u64 *desc;    // (first_word << 24) | n_bases
    const u8 *ascii;    // ASCII DNA or protein residues
    const u64 *aoff;    // ascii offsets [n+1]
    const u64 *adesc;   // tiles of long ASCII sequences: (first_byte << 24) | n_bases; replaces aoff when set
    const u8 *rflags;   // per-read input flags (BSK_ST_HAS_NON_ACGT) or NULL
    u64 n;
    u32 nunits;
    // parameters
    int kind, k, w, s, canonical, circ_ext, m, scale;
    // output
    u64 *refs;     // per read: (first_tuple << 24) | n_tuples, bit 63 (BSK_REF_ROWS): the tuples are 64 apart
    u8 *status;
    u64 *hash;
    u32 *pos;
    u64 cap;  // capacity of hash[]/pos[] in tuples
    // synchronisation / scratch
    u32 *ticket;    // [0] unit ticket, [1] overflow flag, [2..3] the same for a side launch, [4..7] unused since the lists of reads are per workgroup (list_append)
    u32 fixcap;     // entries (reads) of the read list over all segments (list_append): k_syncmer_pk's fixlist, k_minimizer_pk's rlist
    u32 *rlist;     // k_minimizer_pk / k_minimizer_ring: reads for the exact machine (k_minimizer_dense<W, true>): [list_grid] counts, then list_grid segments (list_append)
    u64 *fixlist;   // k_syncmer_pk: the same list (u32 entries) for k_syncmer_fast<W, true>
    u64 *lookback;  // [nunits]
    u64 *total;     // [0] tuples written by the dense (look-back) kernels, [1] overflow-region cursor (slab kernels)
    u64 *ring_h;    // runtime-w ring: per workgroup ring_w*64 entries
    u32 *ring_p;
    u32 ring_w;
    u32 uniform_len;  // != 0: every read has exactly this many bases (synthetic / fixed-length batches)
    u64 slab_read;  // per-sequence slab kernels (protein): tuples reserved per sequence
    u64 list_slab;  // k_minimizer_pkd: the slab of a LISTED read (its list pass runs with slab_read = this)
    u32 unit_rows;  // unit-row kernels (kernels_ring.hpp): rows of 64 tuples in a unit's slab
    u64 ovf_base;   // slab kernels: first tuple index of the overflow region, and its size
    u64 ovf_cap;
    // side launch over a subset of the reads (the reads with a non-ACGT letter of an otherwise 2-bit batch): unit u holds
    // reads subset[64u .. 64u+63]; their tuples go to [out_base, cap)
    const u32 *subset;
    u64 nsub;
    u64 out_base;
    int inplace;  // side launch of a stream kind: every read already owns its line-padded run (refs of the main launch)
    // protein kinds fed with 2-bit DNA (translation fused into the residue fetch, kernels_protein.hpp)
    int frame;       // 1,2,3,-1,-2,-3
    const u8 *lut;   // device codon tables of the context (kernels_translate.hpp layout)
    int pairs;       // KMER, canonical = 0: the alphabet whose PairLetter builds the second strand (bsk_alphabet; 0 = DNAredundant)
    int one_strand;  // KMER, canonical = 0, over tiles: forward codes only (k_two_strand appends the second strand per sequence)
    // length-binned descriptors (bsk_batch::bdesc, k_bin_desc in kernels_host.hpp): the reads of every chunk of 4096 (64 units) are ordered
    // by length class, so that the 64 reads of a unit end together; a binned descriptor carries the read's place in its chunk in bits
    // 12..23 (such batches hold reads of < 4096 bases), and the reference word / status byte of the read a lane holds belong at
    // out_index(), not at unit * 64 + lane
    // two-pass syncmer plan (kernels_syncmer_sel.hpp): selection words [unit][sel_nb][64], per read (offset in its unit << 8 | count),
    // per unit its total and -- after the scan -- where its tuples start
    u32 *sel_mask, *sel_cnt, *sel_utot;
    u64 *sel_ubase, *sel_lookback;
    u32 sel_nb;
    // dense tiles (k_minimizer_pft, kernels_minimizer_pf.hpp): per tile the positions it OWNS, tile-local (lo | hi << 32: kernels_tile.hpp, TileTab::keep),
    // and the position of its first base in its sequence (TileTab::shift); the kernel emits only owned tuples, positions shifted, units packed back
    // to back through `lookback` -- the tile result IS the sequence result, no stitch pass
    const u64 *tkeep, *tshift;
    u32 cls_lo, cls_hi, cls_pretend;  // class plans: see desc_len()
    u32 tk;          // units per ticket of the persistent-wave kernels (0: the kernel's own 4 or 8): a batch with fewer units than the grid has
                     // wavefronts x that number takes smaller tickets -- a pipeline chunk or a class plan's part is latency, not throughput
    u32 list_grid;   // workgroups of the main launch = segments of the list of reads (list_append); the list pass may run with fewer
    u32 len_mask;    // 0xffffff, or 0xfff for binned descriptors
    u32 binned;
};

// The list of READS a packed kernel leaves to the exact machine (k_minimizer_pk / _ring -> k_minimizer_dense<W, true>; k_syncmer_pk ->
// k_syncmer_fast<W, true>).  Every workgroup (= wavefront) of the main kernel owns ONE SEGMENT of it -- list[list_grid + b * seg .. + seg),
// seg = fixcap / list_grid -- filled through a cursor in a register and closed with one store of its count to list[b]: no atomic with
// a return value in the unit loop.  (An `atomicAdd` per unit with a listed read was a load to wait for, and gfx9 counts loads and
// stores in one in-order vmcnt: every such unit waited for the previous copy-out's stores to reach memory -- a batch where 10 % of the
// reads end in a poly-A tail ran the MAIN kernel 59 % slower than a clean one, and k = 31 s = 11 syncmers list a read in half of the
// units.)  The list pass takes the segments (KArgs::list_grid of them) round-robin over its workgroups, 64 reads at a time.
__device__ __forceinline__ void list_append(const KArgs &a, u32 *list, u32 seg, u32 &cur, u64 redo, int lane, u64 r) {
    const u32 at = cur + __builtin_amdgcn_mbcnt_hi((u32)(redo >> 32), __builtin_amdgcn_mbcnt_lo((u32)redo, 0));
    if ((redo >> lane) & 1) {
        if (at < seg) list[a.list_grid + blockIdx.x * seg + at] = (u32)r;
        else atomicOr(&a.ticket[1], 2u);  // the segment is full: the host runs the batch on the 64-bit kernel instead
    }
    cur += (u32)__builtin_popcountll(redo);
}
__device__ __forceinline__ void list_close(u32 *list, u32 seg, u32 cur, int lane) {
    if (lane == 0) list[blockIdx.x] = cur < seg ? cur : seg;
}

// the length a kernel works with.  Class plans (biosketch.hip, run_classed): the kernel of the BULK class runs over the whole batch and a
// read of another class -- length outside [cls_lo, cls_hi] -- pretends cls_pretend bases (the bulk's fixed length, or 0 = an empty SHORT
// entry); what the kernel makes of it is overwritten by the part that owns the read.  cls_hi == 0: no class plan (a scalar branch).
__device__ __forceinline__ u64 desc_len(const KArgs &a, u64 d) {
    u64 L = d & (u64)a.len_mask;
    if (a.cls_hi) L = (L >= (u64)a.cls_lo && L <= (u64)a.cls_hi) ? L : (u64)a.cls_pretend;
    return L;
}
// where the outputs of the read in slot r (descriptor d) go: r itself, or the read's own place in its chunk of 4096
#ifdef BSK_BIN_NOSCATTER  // dev, timing only: what the scattered reference words / status bytes of a binned batch cost
__device__ __forceinline__ u64 out_index(const KArgs &, u64 r, u64) { return r; }
#else
__device__ __forceinline__ u64 out_index(const KArgs &a, u64 r, u64 d) { return a.binned ? ((r & ~4095ULL) | ((d >> 12) & 4095ULL)) : r; }
#endif

// read handled by (unit, lane): the batch position, or the subset entry of a side launch (~0 = no read)
__device__ __forceinline__ u64 read_index(const KArgs &a, u32 unit, int lane) {
    const u64 i = (u64)unit * 64 + lane;
    if (!a.subset) return i;
    return i < a.nsub ? (u64)a.subset[i] : ~0ULL;
}

// byte range of read r in the ASCII buffer
__device__ __forceinline__ void ascii_span(const KArgs &a, u64 r, u64 &off, u64 &L) {
    if (a.adesc) {
        const u64 d = a.adesc[r];
        off = d >> 24;
        L = d & 0xffffffULL;
    } else {
        off = a.aoff[r];
        L = a.aoff[r + 1] - off;
    }
}

// X table: one 16-byte entry per (outgoing code 0..4, incoming code 0..3);
//   .x/.y = rol(seedF[out], k) ^ seedF[in]          (forward strand update)
//   .z/.w = ror(seedR[out], 1) ^ rol(seedR[in], k-1) (reverse strand update)
// outgoing code 4 = "nothing leaves" (warm-up of the first k-1 bases).
__device__ __forceinline__ void build_xtab(uint4 *xt, int k, int tid) {
    if (tid < 20) {
        unsigned out = tid >> 2, in = tid & 3;
        u64 f = seed_fwd_code(in);
        u64 r = rol64(seed_rev_code(in), (unsigned)(k - 1));
        if (out < 4) {
            f ^= rol64(seed_fwd_code(out), (unsigned)k);
            r ^= ror64(seed_rev_code(out), 1);
        }
        xt[tid] = make_uint4((u32)f, (u32)(f >> 32), (u32)r, (u32)(r >> 32));
    }
}
// byte tables for the ASCII path: tin[b] = {seedF[b], rol(seedR[b],k-1)}, tout[b] = {rol(seedF[b],k), ror(seedR[b],1)}
__device__ __forceinline__ void build_bytetabs(uint4 *tin, uint4 *tout, int k, int tid) {
    for (int b = tid; b < 256; b += WAVE) {
        u64 f = seed_fwd_byte(b), r = seed_rev_byte(b);
        u64 fi = f, ri = rol64(r, (unsigned)(k - 1));
        u64 fo = rol64(f, (unsigned)k), ro = ror64(r, 1);
        tin[b] = make_uint4((u32)fi, (u32)(fi >> 32), (u32)ri, (u32)(ri >> 32));
        tout[b] = make_uint4((u32)fo, (u32)(fo >> 32), (u32)ro, (u32)(ro >> 32));
    }
}
__device__ __forceinline__ u64 u64_of(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }

// ---------------------------------------------------------------------------------
// Hash sources.  step(i) returns the hash of k-mer i; i is wave-uniform and is
// called for i = 0,1,2,... in order.  `rev` = 1 iff the reverse-strand hash won.
// ---------------------------------------------------------------------------------
struct NtPacked {  // 2-bit packed input, ACGT only
    const u32 *w;  // this lane's first word
    const uint4 *xt;
    u64 fh, rh;
    u32 win, wout;
    int k, canonical;
    __device__ __forceinline__ void init(const u32 *words, u64 first_word, int k_, int canon, const uint4 *xt_) {
        w = words + first_word;
        xt = xt_;
        k = k_;
        canonical = canon;
        fh = rh = 0;
        win = wout = 0;
        for (int t = 0; t < k - 1; ++t) {  // warm-up: bases 0..k-2 enter, nothing leaves
            if ((t & 15) == 0) win = w[t >> 4];
            uint4 x = xt[16 + ((win >> ((t & 15) * 2)) & 3)];
            fh = rol1(fh) ^ u64_of(x.x, x.y);
            rh = ror1(rh) ^ u64_of(x.z, x.w);
        }
    }
    __device__ __forceinline__ void step(u32 i, u64 &h, u32 &rev) {
        u32 t = i + (u32)k - 1;
        if ((t & 15) == 0) win = w[t >> 4];
        u32 cin = (win >> ((t & 15) * 2)) & 3;
        u32 idx = 16 + cin;
        if (i) {
            u32 p = i - 1;
            if ((p & 15) == 0) wout = w[p >> 4];
            idx = (((wout >> ((p & 15) * 2)) & 3) << 2) | cin;
        }
        uint4 x = xt[idx];
        fh = rol1(fh) ^ u64_of(x.x, x.y);
        rh = ror1(rh) ^ u64_of(x.z, x.w);
        rev = (canonical && rh < fh) ? 1u : 0u;
        h = rev ? rh : fh;
    }
};

struct NtAscii {  // raw bytes, full ntHash-1 seed tables (any byte)
    const u8 *a;
    const uint4 *tin, *tout;
    u64 fh, rh;
    int k, canonical;
    u64 L;  // bytes available (reads are clamped so out-of-range lanes stay in bounds)
    __device__ __forceinline__ void init(const u8 *ascii, u64 off, u64 len, int k_, int canon, const uint4 *tin_,
                                         const uint4 *tout_) {
        a = ascii + off;
        L = len;
        tin = tin_;
        tout = tout_;
        k = k_;
        canonical = canon;
        fh = rh = 0;
        for (int t = 0; t < k - 1; ++t) {
            uint4 x = tin[(u64)t < L ? a[t] : 0];
            fh = rol1(fh) ^ u64_of(x.x, x.y);
            rh = ror1(rh) ^ u64_of(x.z, x.w);
        }
    }
    __device__ __forceinline__ void step(u32 i, u64 &h, u32 &rev) {
        u64 t = (u64)i + (u64)k - 1;
        uint4 x = tin[t < L ? a[t] : 0];
        fh = rol1(fh) ^ u64_of(x.x, x.y);
        rh = ror1(rh) ^ u64_of(x.z, x.w);
        if (i) {
            uint4 y = tout[(u64)(i - 1) < L ? a[i - 1] : 0];
            fh ^= u64_of(y.x, y.y);
            rh ^= u64_of(y.z, y.w);
        }
        rev = (canonical && rh < fh) ? 1u : 0u;
        h = rev ? rh : fh;
    }
};

// ---------------------------------------------------------------------------------
// Per-lane LDS staging of selected tuples.  Slot of tuple e of lane l:
//   e*64 + ((l + e) & 63)
// -> writes (many lanes, similar e) and the copy-out reads (one lane's run of
//    consecutive e) are both bank-conflict free.
// ---------------------------------------------------------------------------------
template <int CAP>
struct Stage {
    u64 *sh;    // [CAP*64]
    u32 *sp;    // [CAP*64]
    u16 *smap;  // [CAP*64] wave-relative output index -> slot
    __device__ __forceinline__ static u32 slot(u32 e, int lane) { return e * 64u + ((u32)(lane + e) & 63u); }
};

// Window minimizer over a hash source: the closed form of NextMinimizer
// (sketch.go:205-309; DESIGN.md "closed form"): per window of W
// consecutive k-mers the LEFTMOST minimum, emitted when its position changes.
// Sliding minimum by the two-pass block decomposition: blocks of W k-mers; P =
// running prefix minimum of the current block, ring[] = suffix minima of the
// previous block; window min = min(ring[o+1], P) with the older element winning ties.
// The ring lives in global scratch (layout [slot][lane], so every access is one
// coalesced 512-byte line per wave) because W is a run-time value here.
template <class Src, int CAP, bool DIRECT>
__device__ __forceinline__ void window_pass(Src &src, u32 nk, u32 nk_max, int W, u64 *ring_h, u32 *ring_p, int lane,
                                            Stage<CAP> st, u32 &cnt, u32 &tie, u64 *ghash, u32 *gpos, u64 gbase) {
    u64 Ph = 0;
    u32 Pp = 0, prev = 0xffffffffu;
    int o = 0;
    bool first = true;
    for (u32 i = 0; i < nk_max; ++i) {
        u64 h;
        u32 rev;
        src.step(i, h, rev);
        const u32 ps = i | (rev << 31);
        const bool act = i < nk;
        if (o == 0 || h < Ph) {
            Ph = h;
            Pp = ps;
        }
        if (!first || o == W - 1) {
            u64 mh = Ph;
            u32 mp = Pp;
            if (o != W - 1) {
                u64 Sh = ring_h[(o + 1) * 64 + lane];
                u32 Sp = ring_p[(o + 1) * 64 + lane];
                if (!(Ph < Sh)) {
                    mh = Sh;
                    mp = Sp;
                }
            }
            const bool emit = act && mp != prev;
            prev = mp;
            if (emit) {
                if (!DIRECT) {
                    if (cnt < (u32)CAP) {
                        u32 sl = Stage<CAP>::slot(cnt, lane);
                        st.sh[sl] = mh;
                        st.sp[sl] = mp;
                    }
                } else {
                    ghash[gbase + cnt] = mh;
                    gpos[gbase + cnt] = mp;
                }
                cnt++;
            }
        }
        ring_h[o * 64 + lane] = h;
        ring_p[o * 64 + lane] = ps;
        if (o == W - 1) {
            u64 nh = h;
            u32 np = ps;
            u32 dup = 0;  // BSK_ST_FIRST_WINDOW_TIE: the minimum of [q, W) occurs twice for some q (kernels_fast.hpp, suffix_min_pass)
            for (int q = W - 2; q >= 0; --q) {  // ring[q] = min(ring[q..W-1]), leftmost on ties
                u64 ah = ring_h[q * 64 + lane];
                u32 ap = ring_p[q * 64 + lane];
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

// Unit epilogue shared by every tuple-producing kernel: wave scan of the per-lane
// counts, look-back for the unit's global base, LDS -> HBM copy-out in read order,
// per-read references, status bytes.  Returns the unit's global base and whether the
// result buffer is too small (then nothing is written, only refs/total).
template <int CAP>
__device__ __forceinline__ u64 unit_epilogue(const KArgs &a, u32 unit, int lane, u64 r, u32 c, Stage<CAP> st,
                                             u32 &excl_out, bool &ovf_out) {
    const u32 incl = wave_incl_scan_u32(c, lane);
    const u32 excl = incl - c;
    const u32 T = wave_bcast_u32(incl, 63);
    const u64 base = a.out_base + lookback_exclusive(a.lookback, unit, (u64)T, lane);
    const bool ovf = base + T > a.cap;
    // A lane that selected more than CAP tuples could not stage them all: then the whole unit is
    // written by the DIRECT re-run instead (rare; the caller checks the same ballot).
    const bool any_over = __ballot(c > (u32)CAP) != 0;
    if (!ovf && !any_over) {
        // smap: wave-relative output index -> LDS slot
        const u32 cmax = wave_max_u32(c);
        for (u32 e = 0; e < cmax; ++e)
            if (e < c) st.smap[excl + e] = (u16)Stage<CAP>::slot(e, lane);
        wave_sync_lds();
        for (u32 t = lane; t < T; t += 64) {
            const u32 sl = st.smap[t];
            a.hash[base + t] = st.sh[sl];
            if (a.pos) a.pos[base + t] = st.sp[sl];
        }
        wave_sync_lds();
    } else if (!ovf) {
        // nothing: DIRECT pass follows
    } else if (lane == 0) {
        atomicOr(&a.ticket[1], 1u);
    }
    if (r < a.n) a.refs[r] = ((base + excl) << 24) | c;
    if (unit == a.nunits - 1 && lane == 63) *a.total = base + incl;
    excl_out = excl;
    ovf_out = ovf;
    return base;
}

#define BSK_GEN_CAP 32

// ---- MINIMIZER, generic (kind BSK_MINIMIZER; any k, any w; packed or ASCII) ----
template <int ENC>  // 0 packed, 1 ascii
__global__ __launch_bounds__(64) void k_minimizer_generic(KArgs a) {
    constexpr int CAP = BSK_GEN_CAP;
    __shared__ uint4 s_tab[ENC ? 512 : 32];
    __shared__ u64 s_h[CAP * 64];
    __shared__ u32 s_p[CAP * 64];
    __shared__ u16 s_m[CAP * 64];
    const int lane = lane_id();
    if (ENC) build_bytetabs(s_tab, s_tab + 256, a.k, lane);
    else build_xtab(s_tab, a.k, lane);
    __syncthreads();
    Stage<CAP> st{s_h, s_p, s_m};
    u64 *ring_h = a.ring_h + (u64)blockIdx.x * a.ring_w * 64;
    u32 *ring_p = a.ring_p + (u64)blockIdx.x * a.ring_w * 64;
    const int W = a.w;
    for (;;) {
        const u32 unit = next_ticket(a.ticket, lane);
        if (unit >= a.nunits) break;
        const u64 r = read_index(a, unit, lane);
        u64 off = 0, L = 0;
        if (r < a.n) {
            if (ENC) {
                ascii_span(a, r, off, L);
            } else {
                u64 d = a.desc[r];
                off = d >> 24;
                L = d & 0xffffffULL;
            }
        }
        // NewMinimizerSketch sketch.go:92: len(S.Seq) < k+w-1 -> ErrShortSeq (on the un-extended length)
        const bool ok = r < a.n && (L - (u64)a.circ_ext) + 1 >= (u64)a.k + (u64)W && L >= (u64)a.circ_ext;
        const u32 nk = ok ? (u32)(L - a.k + 1) : 0u;
        const u32 nk_max = wave_max_u32(nk);
        u32 cnt = 0, tie = 0;
        if (nk_max) {
            if (ENC) {
                NtAscii src;
                src.init(a.ascii, off, L, a.k, 1, s_tab, s_tab + 256);
                window_pass<NtAscii, CAP, false>(src, nk, nk_max, W, ring_h, ring_p, lane, st, cnt, tie, a.hash, a.pos, 0);
            } else {
                NtPacked src;
                src.init(a.words, off, a.k, 1, s_tab);
                window_pass<NtPacked, CAP, false>(src, nk, nk_max, W, ring_h, ring_p, lane, st, cnt, tie, a.hash, a.pos, 0);
            }
        }
        u32 excl;
        bool ovf;
        const u64 base = unit_epilogue<CAP>(a, unit, lane, r, cnt, st, excl, ovf);
        if (!ovf && __ballot(cnt > (u32)CAP)) {  // rare: a lane selected more than CAP tuples -> recompute, write straight to HBM
            u32 cnt2 = 0, tie2 = 0;
            if (ENC) {
                NtAscii src;
                src.init(a.ascii, off, L, a.k, 1, s_tab, s_tab + 256);
                window_pass<NtAscii, CAP, true>(src, nk, nk_max, W, ring_h, ring_p, lane, st, cnt2, tie2, a.hash, a.pos, base + excl);
            } else {
                NtPacked src;
                src.init(a.words, off, a.k, 1, s_tab);
                window_pass<NtPacked, CAP, true>(src, nk, nk_max, W, ring_h, ring_p, lane, st, cnt2, tie2, a.hash, a.pos, base + excl);
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

// ---------------------------------------------------------------------------------
// "Every position" kinds: value i of read r goes to out[first_tuple(r) + i].  Counts are
// known from the lengths alone, so the look-back runs BEFORE the hashing and values
// stream out through a 64x16 LDS transpose tile (row = read): each flush writes
// 128 contiguous bytes per read with 16-byte stores.
// ---------------------------------------------------------------------------------
#define TILE_LD 18  // u64 per row: 16 + 2 pad (16-byte aligned rows)
struct __attribute__((aligned(8))) u64x2_a8 {
    u64 a, b;
};

template <int ENC>
__global__ __launch_bounds__(64) void k_nthash_stream(KArgs a) {
    __shared__ uint4 s_tab[ENC ? 512 : 32];
    __shared__ u64 s_tile[64 * TILE_LD];
    __shared__ u64 s_off[64];
    __shared__ u32 s_nk[64];
    const int lane = lane_id();
    if (ENC) build_bytetabs(s_tab, s_tab + 256, a.k, lane);
    else build_xtab(s_tab, a.k, lane);
    __syncthreads();
    for (;;) {
        const u32 unit = next_ticket(a.ticket, lane);
        if (unit >= a.nunits) break;
        const u64 r = read_index(a, unit, lane);
        u64 off = 0, L = 0;
        if (r < a.n) {
            if (ENC) {
                ascii_span(a, r, off, L);
            } else {
                u64 d = a.desc[r];
                off = d >> 24;
                L = d & 0xffffffULL;
            }
        }
        // NewHashIterator iterator.go:619: len(s.Seq) < k -> ErrShortSeq
        const bool ok = r < a.n && L >= (u64)a.circ_ext && (L - (u64)a.circ_ext) >= (u64)a.k;
        const u32 nk = ok ? (u32)(L - a.k + 1) : 0u;
        const u32 nk_max = wave_max_u32(nk);
        const u64 incl = wave_incl_scan_u64((u64)nk, lane);
        const u64 T = wave_bcast_u64(incl, 63);
        // fixed-length batch: every earlier unit holds exactly 64*nk values, no prefix chain needed
        u64 first = 0;
        bool ovf = false;
        if (a.inplace) {
            if (r < a.n) first = a.refs[r] >> 24;
        } else {
            const u64 base = a.out_base + (a.uniform_len ? (u64)unit * 64 * nk_max : lookback_exclusive(a.lookback, unit, T, lane));
            ovf = base + T > a.cap;
            if (ovf && lane == 0) atomicOr(&a.ticket[1], 1u);
            if (unit == a.nunits - 1 && lane == 63) *a.total = base + incl;
            first = base + incl - nk;
        }
        if (r < a.n) a.refs[r] = (first << 24) | nk;
        if (r < a.n) {
            u8 sbyte = ok ? BSK_ST_OK : BSK_ST_SHORT;
            if (ok && a.rflags) sbyte |= a.rflags[r];
            a.status[r] = sbyte;
        }
        if (ovf || nk_max == 0) continue;
        s_off[lane] = first;
        s_nk[lane] = nk;
        NtPacked sp;
        NtAscii sa;
        if (ENC) sa.init(a.ascii, off, L, a.k, a.canonical, s_tab, s_tab + 256);
        else sp.init(a.words, off, a.k, a.canonical, s_tab);
        wave_sync_lds();
        for (u32 i = 0; i < nk_max; ++i) {
            u64 h;
            u32 rev;
            if (ENC) sa.step(i, h, rev);
            else sp.step(i, h, rev);
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
                    const u64 v0 = s_tile[row * TILE_LD + col], v1 = s_tile[row * TILE_LD + col + 1];
                    if (ia + 1 < nkr) {
                        // two u64 = one 16-byte store (dst is only 8-byte aligned: dword alignment is
                        // enough for global_store_dwordx4)
                        u64x2_a8 vv;
                        vv.a = v0;
                        vv.b = v1;
                        *reinterpret_cast<u64x2_a8 *>(dst) = vv;
                    } else if (ia < nkr) {
                        dst[0] = v0;
                    }
                }
                wave_sync_lds();
            }
        }
    }
}

}  // namespace bsk
