// kernels_pk.hpp -- k_minimizer_pk<W>: the minimizer kernel for W <= 13 and 2-bit reads shorter than 32 768 bases.
//
// Same mapping, rolling hash, paired LDS staging columns and slab output as k_minimizer_fast (kernels_fast.hpp).  What differs is
// the window machine (NextMinimizer, sketches/sketch.go:205-309 -- closed form: leftmost argmin of every window, emitted when it
// changes):
//   * an element of the window is ONE 32-bit word  key | idx : the upper 27 bits of the canonical hash and a 5-bit slot number
//     (block parity * 16 + offset in the block).  Prefix minimum, suffix minimum and their combination are then one full-rate
//     v_min_u32 each instead of a 64-bit compare and three v_cndmask (hash lo, hash hi, position): 3 VOP2 against 12 VOP3 per step.
//   * what a window selected is recorded as a bit: bm |= 1 << idx (one v_lshl_or_b32; setting a bit twice IS the reference's
//     "emit only when the position changed").  Slot o of the previous block is final when step o of the current block begins, so
//     the staging store of a step writes STATIC registers (that slot's hash pair and strand) and the staging pointer advances by
//     bit o of bm (v_bfe + v_mad_i24) -- no selects, no position compare, no running position register.
//   * exactness.  The 64-bit leftmost minimum of a window is the packed minimum unless two elements of the window share the
//     minimal 27-bit key.  Every such pair meets in one of the three min operations (both in the current block: when the later one
//     meets the prefix minimum; both in the previous block: in the suffix pass at the earlier one; one in each: where prefix and
//     suffix minima are combined -- DESIGN.md), and a key tie there is (a ^ b) < 32: the kernel keeps the minimum of those
//     xors (two full-rate ops per min operation).  A READ that saw one -- accidental key ties are 6e-4 of the units on random reads;
//     reads with real 64-bit ties (a k-mer and its reverse complement in one window, homopolymers) always -- is appended to a list
//     and run afterwards by the exact 64-bit machine (k_minimizer_dense<W, true>), which also evaluates BSK_ST_FIRST_WINDOW_TIE;
//     everywhere else no tie exists and the flag is 0.
// LDS layout: paired columns (lanes l and l + 32 fill one column from both ends) and 16-bit positions as in k_minimizer_fast, 58 rows,
// eight waves per CU (PkLds); the copy-out is pk_copyout below.  The kernel is bound by what two waves per SIMD issue in order: what
// pays is removing instructions of any kind, not cheaper encodings (profiles/NOTEBOOK.md, round 3).
// (Tried first: byte positions + the copy-out's tables laid over the hash tables = 17.7 KB, nine waves per CU under a 168-VGPR cap.
// The ninth wave was worth 3-4.7 %; the cap cost spills, the byte positions a wrap rule in the copy-out and a read-length limit.)
#pragma once
#include <utility>

#include "kernels_fast.hpp"

namespace bsk {

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): an unrolled loop whose counter is a constant expression
// (inline-asm immediates)
template <int... Is, class F>
__device__ __forceinline__ void pk_unroll_impl(std::integer_sequence<int, Is...>, F &&f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void pk_unroll(F &&f) {
    pk_unroll_impl(std::make_integer_sequence<int, N>{}, f);
}

// LDS plan of one wavefront.  Paired columns as PLds (kernels_fast.hpp), but 58 rows instead of 56: a column fills up when its two
// reads select 58 tuples together -- 0.8 % of the units at k=21 w=11, 150 bp, against 4.1 % with 56 rows (measured: the tail is
// heavier than normal), and both reads of such a column go to the exact machine afterwards.  The room comes from laying the copy-out's tables
// (head words, owner table) over the two hash-phase tables, which a wavefront keeps in four VGPRs and writes back at the start of
// every unit (PkTabs).  20 248 B: still eight waves per CU.
#ifndef BSK_PK_ROWS
#define BSK_PK_ROWS 58
#endif
struct PkLds {
    static constexpr int PR = BSK_PK_ROWS;
    static constexpr int ROW = 33;      // 32 columns + 1: consecutive rows of a column rotate through the banks
    static constexpr int TAB = 0;       // 20 x uint4 update table           } hash phase
    static constexpr int TAB2 = 320;    // 16 x uint4 two-base warm-up table }
    static constexpr int NHEADS = (32 * (PR - 1)) / 64 + 2;
    static constexpr int HEADS = 0;     // u64 [NHEADS]            } copy-out
    static constexpr int CTAB = 256;    // u64 [64]: owner table   }
    static constexpr int SH = 768;                                          // u64 [(PR+1)*33]
    static constexpr int SP = SH + (PR + 1) * ROW * 8;                      // u16 [(PR+1)*33]
    static constexpr int TOTAL = (SP + (PR + 1) * ROW * 2 + 15) & ~15;
    static_assert(NHEADS * 8 <= CTAB && CTAB + 512 <= SH && TAB2 + 256 <= SH && TOTAL <= 20480, "PkLds");
};

// the two hash-phase tables, one row per lane (lanes 0..19: update table, lanes 32..47: two-base warm-up table)
struct PkTabs {
    u32x4 row;
    u32 at;  // LDS offset of the lane's row, or ~0
    __device__ __forceinline__ void init(int k, int lane, u32 tab = (u32)PkLds::TAB, u32 tab2 = (u32)PkLds::TAB2) {
        row = (u32x4){0, 0, 0, 0};
        at = 0xffffffffu;
        if (lane < 20) {  // build_xtab's rows
            const unsigned out = (unsigned)lane >> 2, in = (unsigned)lane & 3u;
            u64 f = seed_fwd_code(in);
            u64 r = rol64(seed_rev_code(in), (unsigned)(k - 1));
            if (out < 4) {
                f ^= rol64(seed_fwd_code(out), (unsigned)k);
                r ^= ror64(seed_rev_code(out), 1);
            }
            row = (u32x4){(u32)f, (u32)(f >> 32), (u32)r, (u32)(r >> 32)};
            at = tab + (u32)lane * 16u;
        } else if (lane >= 32 && lane < 48) {  // entry (c0 | c1 << 2), c0 entering first, nothing leaving
            const unsigned c0 = (unsigned)lane & 3u, c1 = ((unsigned)lane >> 2) & 3u;
            const u64 f = rol64(seed_fwd_code(c0), 1) ^ seed_fwd_code(c1);
            const u64 r = ror64(rol64(seed_rev_code(c0), (unsigned)(k - 1)), 1) ^ rol64(seed_rev_code(c1), (unsigned)(k - 1));
            row = (u32x4){(u32)f, (u32)(f >> 32), (u32)r, (u32)(r >> 32)};
            at = tab2 + (u32)(lane - 32) * 16u;
        }
    }
    __device__ __forceinline__ void write(LDSQ char *ldsq) const {
        if (at != 0xffffffffu) *reinterpret_cast<LDSQ u32x4 *>(ldsq + at) = row;
        wave_sync_lds();
    }
};

template <int W>
struct PkCfg {
    static constexpr int XC = W > 12 ? 4 : W;  // table rows fetched per chunk (all up front for W <= 12)
};

#define PKNW 16
typedef u32 u32x16 __attribute__((ext_vector_type(16)));
struct PkWords {
    u32x4 a, b, c, d;  // the first PKNW packed words of a read
};
// Loads the s_waitcnt pass does not see (see PkMin::word2): the caller waits with pk_wait_loads() before the first use.
__device__ __forceinline__ PkWords pk_load_words(const u32 *p) {
    PkWords r;
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
                 "global_load_dwordx4 %2, %4, off offset:32\n\tglobal_load_dwordx4 %3, %4, off offset:48"
                 : "=&v"(r.a), "=&v"(r.b), "=&v"(r.c), "=&v"(r.d)
                 : "v"(p));
    return r;
}
__device__ __forceinline__ u64 pk_load_u64(const u64 *p) {
    u64 r;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(r) : "v"(p));
    return r;
}
// The wait "produces" the loaded values: anything computed from them is thereby ordered behind it (a bare asm volatile is not a
// barrier for the arithmetic on its neighbours' results -- the address of the words was computed from the descriptor ahead of it).
__device__ __forceinline__ void pk_wait_loads(u64 &d0, u64 &d1) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(d0), "+v"(d1)::"memory"); }
__device__ __forceinline__ void pk_wait_loads(PkWords &p) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(p.a), "+v"(p.b), "+v"(p.c), "+v"(p.d)::"memory"); }
__device__ __forceinline__ void pk_wait_loads(PkWords &p, u64 &d0, u32 &f) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(p.a), "+v"(p.b), "+v"(p.c), "+v"(p.d), "+v"(d0), "+v"(f)::"memory");
}
__device__ __forceinline__ u32 pk_load_u8(const u8 *p) {
    u32 r;
    asm volatile("global_load_ubyte %0, %1, off" : "=&v"(r) : "v"(p));
    return r;
}
// the packed window machine + staging of one read per lane
// LONG: reads of more than 16 (PKNW - 1) bases (their further words are loaded inside the k-mer loop)
// RINGM (k_minimizer_pkd, kernels_pkd.hpp): the same machine over FLds staging -- the lane's CAP + 1 rows are a ring that the kernel
// flushes to the read's own slab every few blocks (flush_groups), so a read may select any number of tuples; the words of a block are
// requested one block ahead (in_lo .. out_hi) and there is no register copy of the read, no paired column and no guard.
// SELM (k_minimizer_pft, kernels_minimizer_pf.hpp, round 6): the SELECTION alone -- nothing is staged: the finished block's selection bits
// (bit o = k-mer o of the block) go to row (block number) of LY::MASK, one word per lane, and the unit's own emit phase hashes what was selected.
template <int W, bool LONG, class LY_ = PkLds, bool RINGM = false, bool SELM = false>
struct PkMin {
    typedef LY_ LY;
    const u32 *__restrict__ w;
    LDSQ char *lds;
    int k, lane;
    u32 nk;
    u32 fl, fh_, rl, rh_;
    u32 S[W];   // packed suffix minima of the previous block (then raw packed values of the current one)
    u64 H[W];   // canonical hashes of the previous block, slot by slot replaced by the current block's
    u32 SB[W];  // strand << 15 of the same slots
    u32 P, bm, tmin;
    u32 nsel;  // SELM: the lane's selections so far
    u32 slot, spare;
    u32 glo, gspan;  // guard(): the lane's staging pointer may start a block in [glo, glo + gspan]
    u32 c8000;
    int sstep;
    u32 send;                          // RINGM: the slot one row past the lane's last row
    u32 in_lo, in_hi, out_lo, out_hi;  // RINGM: the packed words of the next block to run
    u32x4 pw;                          // RINGM: the read's first four words
    // The read's first PKNW packed words (240 bases) live in registers, loaded one unit ahead, BEFORE the previous unit's copy-out
    // stores: gfx9 counts loads and stores in one in-order vmcnt, so a load issued inside the k-mer loop waits for every copy-out
    // store queued before it.  A word is picked by its wave-uniform index (s_set_gpr_idx + v_mov); longer reads load the rest.
    u32x16 wr;

    __device__ __forceinline__ void set_words(const PkWords &p) {
        wr = (u32x16){p.a.x, p.a.y, p.a.z, p.a.w, p.b.x, p.b.y, p.b.z, p.b.w, p.c.x, p.c.y, p.c.z, p.c.w, p.d.x, p.d.y, p.d.z, p.d.w};
    }
    // Words i and i + 1 of the read, i wave-uniform: register-indexed moves (s_set_gpr_idx + v_mov).  (One branch per index with static
    // register reads instead: 7 % slower -- the branches cut the block's head into pieces.)  The compiler's s_waitcnt pass cannot tell
    // which register such a move reads and waits for EVERY load it knows to be in flight: the loads of the next unit's words are
    // therefore issued from inline asm, which it does not see (pk_load_words), and waited for by hand.
    __device__ __forceinline__ void word2(u32 i, u32 &lo, u32 &hi) const {
        const u32 iu = (u32)__builtin_amdgcn_readfirstlane((int)i);
        if (!LONG || iu + 1 < (u32)PKNW) {
            lo = wr[iu], hi = wr[iu + 1];
        } else {
            lo = w[iu], hi = w[iu + 1];
        }
    }
    __device__ __forceinline__ u32 word(u32 i) const {
        const u32 iu = (u32)__builtin_amdgcn_readfirstlane((int)i);
        if constexpr (RINGM) {  // the first four words came with the unit (requested a unit ahead), the rest is loaded here
            if (iu < 4u) return iu == 0 ? pw.x : iu == 1 ? pw.y : iu == 2 ? pw.z : pw.w;
            return w[iu];
        }
        if (!LONG || iu < (u32)PKNW) return wr[iu];
        return w[iu];
    }
    // RINGM: what block i0 reads (FastMin::load_block_words).  Block 0's come through word() -- the unit's first four words or a load --,
    // every other block's straight from memory: a branch per word cut the head of the block into pieces (7 % of k_minimizer_pk, 9 % here)
    __device__ __forceinline__ void load_block_words(u32 i0) {
        const u32 t0 = i0 + (u32)k - 1, p0 = i0 - 1;
        in_lo = w[t0 >> 4];
        in_hi = w[(t0 >> 4) + 1];
        out_lo = w[p0 >> 4];
        out_hi = w[(p0 >> 4) + 1];
    }
    __device__ __forceinline__ void load_block0_words() {
        const u32 t0 = (u32)k - 1;
        in_lo = word(t0 >> 4);
        in_hi = word((t0 >> 4) + 1);
        out_lo = word(0);
        out_hi = word(1);
    }
    __device__ __forceinline__ void roll(u32x4 x) {
        const u32 a = __builtin_amdgcn_alignbit(fl, fh_, 31), b = __builtin_amdgcn_alignbit(fh_, fl, 31);
        const u32 c = __builtin_amdgcn_alignbit(rh_, rl, 1), d = __builtin_amdgcn_alignbit(rl, rh_, 1);
        fl = a ^ x.x;
        fh_ = b ^ x.y;
        rl = c ^ x.z;
        rh_ = d ^ x.w;
    }
    __device__ __forceinline__ void roll2(u32x4 x) {
        const u32 a = __builtin_amdgcn_alignbit(fl, fh_, 30), b = __builtin_amdgcn_alignbit(fh_, fl, 30);
        const u32 c = __builtin_amdgcn_alignbit(rh_, rl, 2), d = __builtin_amdgcn_alignbit(rl, rh_, 2);
        fl = a ^ x.x;
        fh_ = b ^ x.y;
        rl = c ^ x.z;
        rh_ = d ^ x.w;
    }

    // Before the W staging steps of a block: a lane whose pointer could leave its column during them (a low lane within W - 1 rows of
    // the spare row, a high lane within W - 1 rows of row 0: alone more tuples than a column holds with its partner's) is parked on the
    // spare row for the rest of the unit -- its count then reads as a full column and both reads of the column go to the list.  Four
    // instructions per block instead of a clamp (v_min_u32) in every staging step.
    __device__ __forceinline__ void guard() {
        if constexpr (RINGM || SELM) return;
        const bool in = slot - glo <= gspan;
        slot = in ? slot : spare;
        sstep = in ? sstep : 0;
    }
    // one staging step: slot o of the block whose slot 0 is k-mer pbase (wave-uniform)
    template <int IDX, int O>
    __device__ __forceinline__ void emit(u32 pbase) {
        if constexpr (!SELM) emit_staged<IDX, O>(pbase);
    }
    template <int IDX, int O>
    __device__ __forceinline__ void emit_staged(u32 pbase) {
        const u32 b = (bm >> IDX) & 1u;  // v_bfe_u32
        u32 pv;  // strand << 15 | position: one v_add3_u32 (scalar base, inline slot number); as C it is an s_add per step and a v_or
        asm("v_add3_u32 %0, %1, %2, %3" : "=v"(pv) : "v"(SB[O]), "s"(pbase), "n"(O));
#if defined(PK_MASKST)  // dev: the staging writes under the selection bit's lane mask (about a sixth of the lanes), no branch
        {
            const u32 a0 = (u32)(uintptr_t)(lds + LY::SH) + slot, a1 = (u32)(uintptr_t)(lds + LY::SP) + (slot >> 2);
            u64 sv;
            asm volatile("v_cmp_ne_u32_e32 vcc, 0, %1\n\ts_and_saveexec_b64 %0, vcc\n\tds_write_b64 %2, %3\n\tds_write_b16 %4, %5\n\ts_mov_b64 exec, %0"
                         : "=&s"(sv) : "v"(b), "v"(a0), "v"(H[O]), "v"(a1), "v"(pv) : "vcc", "memory");
        }
#elif defined(PK_NOPOS)  // dev (round 6, timing only): hashes-only staging -- the per-step position write and its v_add3 gone, nothing rebuilds the positions:
        // an UPPER BOUND of what any exact form of it can gain (the review's experiment (a); PK_STRMASK adds the strand mask an exact form needs per step)
        *reinterpret_cast<LDSQ u64 *>(lds + LY::SH + slot) = H[O];
#ifdef PK_STRMASK
        asm volatile("" ::"v"(pv));
#endif
#elif !defined(PK_NOSTAGE)  // (dev knock-outs, timing only: PK_NOSTAGE, PK_NOTIE, PK_NOTAB, PK_NOCOPY)
        *reinterpret_cast<LDSQ u64 *>(lds + LY::SH + slot) = H[O];
        *reinterpret_cast<LDSQ u16 *>(lds + LY::SP + (slot >> 2)) = (u16)pv;
#else
        asm volatile("" ::"v"(H[O]), "v"(pv), "v"(slot));
#endif
        asm("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(slot) : "v"(b), "v"(sstep));  // (the compiler's bfe_i32 + and + add is one instruction more)
        if constexpr (RINGM) slot = slot == send ? (u32)lane * 8u : slot;  // the ring's wrap
    }

    // FIRST: block 0 (nothing leaves at slot 0, no window is complete before its last step, nothing to emit; okbit = 0 for lanes
    //        without a read)
    // RAG:   windows end per lane (ragged batch, or the wave's last, partial block)
    // PAR:   parity of the block: its slots are idx PAR*16 + o, the previous block's (1-PAR)*16 + o
    // SUFFIX: another block follows (the suffix minima are needed)
    template <bool FIRST, bool RAG, int PAR>
    __device__ __forceinline__ void block(u32 i0, u32 okbit, bool suffix) {
        constexpr int CB = PAR * 16, PB = (1 - PAR) * 16;
        constexpr int XC = PkCfg<W>::XC;
        const u32 t0 = i0 + (u32)k - 1, p0 = FIRST ? 0u : i0 - 1u;
        u32 in_lo, in_hi, out_lo, out_hi;
        if constexpr (RINGM) {  // requested a block ago; the next block's go out now
            in_lo = this->in_lo, in_hi = this->in_hi, out_lo = this->out_lo, out_hi = this->out_hi;
            load_block_words(i0 + (u32)W);
        } else {
            word2(t0 >> 4, in_lo, in_hi);
            word2(p0 >> 4, out_lo, out_hi);
        }
        const u32 cinb = __builtin_amdgcn_alignbit(in_hi, in_lo, (t0 & 15) * 2);  // code of slot o at bits [2o, 2o+2)
        u32 coutb;
        if (FIRST) coutb = out_lo << 2;  // slot 0: nothing leaves; slot o >= 1 sees base o-1
        else coutb = __builtin_amdgcn_alignbit(out_hi, out_lo, (p0 & 15) * 2);
        // table offsets: nibble j of E / O = (out << 2 | in) of slot 2j / 2j+1, so a slot's row offset is (word >> n) & 0xF0
        const u32 E = (cinb & 0x33333333u) | ((coutb & 0x33333333u) << 2);
        const u32 O = ((cinb >> 2) & 0x33333333u) | (coutb & 0xCCCCCCCCu);
        // nibble j times 16 = the high nibble of byte (j - 1) / 2 of the word (j odd) or of byte j / 2 of the word << 4 (j even): ONE
        // v_and_b32_sdwa with a byte select instead of a shift and an and per slot
        const u32 E4 = E << 4, O4 = O << 4;
        u32x4 xs[W];
        auto fetch = [&](int o0) {
#pragma unroll
            for (int o = o0; o < o0 + XC && o < W; ++o) {
                const int j = o >> 1;
                const u32 src = (j & 1) ? ((o & 1) ? O : E) : ((o & 1) ? O4 : E4);
                u32 a;
                switch (j >> 1) {
                    case 0: asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(a) : "v"(src), "s"(0xF0u)); break;
                    case 1: asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(a) : "v"(src), "s"(0xF0u)); break;
                    case 2: asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(a) : "v"(src), "s"(0xF0u)); break;
                    default: asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(a) : "v"(src), "s"(0xF0u)); break;
                }
                if (FIRST && o == 0) a = 0x100u | (a & 0x30u);  // row "nothing leaves"
#ifndef PK_NOTAB
                xs[o] = *reinterpret_cast<LDSQ const u32x4 *>(lds + LY::TAB + a);
#else
                xs[o] = (u32x4){a, a ^ E, a + O, a ^ cinb};
#endif
            }
        };
        fetch(0);
        if (XC < W) fetch(XC);
        const u32 pbase = (u32)__builtin_amdgcn_readfirstlane((int)(i0 - (u32)W));
        if (!FIRST) guard();
        u32 vb = 0;
        if (RAG && !FIRST) {  // bit o: the window ending at slot o exists for this lane
            const int left = (int)nk - (int)i0;
            const u32 nv = (u32)(left < 0 ? 0 : left > W ? W : left);
            vb = (1u << nv) - 1u;
        }
        pk_unroll<W>([&](auto oc) {
            constexpr int o = decltype(oc)::value;
            if (XC < W && o && o % XC == 0 && o + XC < W) {
                __builtin_amdgcn_sched_barrier(0);
                fetch(o + XC);
            }
            roll(xs[o]);
            const lmask rev = lt64(rl, rh_, fl, fh_);
            if (!FIRST) this->template emit<PB + o, o>(pbase);  // the previous block's slot o, before its registers are re-used
            const u32 hl = sel(rev, rl, fl), hh = sel(rev, rh_, fh_);
            H[o] = ((u64)hh << 32) | hl;
            asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(SB[o]) : "v"(c8000), "s"(rev));  // strand << 15 in one select (0x8000 kept in a VGPR: VOP3 takes no literal)
            u32 pk;  // (hh & ~31) | idx: one v_and_or_b32 with the mask in an SGPR (VOP3 takes no literal on gfx9: the compiler's form is and + or)
            asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(pk) : "v"(hh), "s"(0xffffffe0u), "n"(CB + o));
            if (o == 0) {
                P = pk;
            } else {
#ifndef PK_NOTIE
                const u32 d = pk ^ P;
                tmin = tmin < d ? tmin : d;
#endif
                P = P < pk ? P : pk;
            }
            if (!FIRST || o == W - 1) {
                u32 m = P;
                if constexpr (o != W - 1) {
#ifndef PK_NOTIE
                    const u32 d = P ^ S[o + 1];
                    tmin = tmin < d ? tmin : d;
#endif
                    m = P < S[o + 1] ? P : S[o + 1];
                }
                u32 one = 1u;
                if (FIRST) one = okbit;
                else if (RAG) one = (vb >> o) & 1u;
                bm |= one << (m & 31u);  // v_lshl_or_b32
            }
            S[o] = pk;
        });
        if (suffix) {
#pragma unroll
            for (int q = W - 2; q >= 0; --q) {
#ifndef PK_NOTIE
                const u32 d = S[q] ^ S[q + 1];
                tmin = tmin < d ? tmin : d;
#endif
                S[q] = S[q] < S[q + 1] ? S[q] : S[q + 1];
            }
        }
        if constexpr (SELM) {
            if (!FIRST) {  // the previous block (number i0 / W - 1) is final: its word leaves
                const u32 wsel = (bm >> PB) & ((1u << W) - 1u);
                *reinterpret_cast<LDSQ u32 *>(lds + LY::MASK + (i0 / (u32)W - 1u) * 256u + (u32)lane * 4u) = wsel;
                nsel += (u32)__builtin_popcount(wsel);
            }
        }
        if (!FIRST) bm &= PAR ? 0xffff0000u : 0x0000ffffu;  // the previous block's slots are all emitted
    }

    // the last block's own slots
    template <int PAR>
    __device__ __forceinline__ void drain(u32 i0) {
        if constexpr (SELM) {  // the last block's own word
            const u32 wsel = (bm >> (PAR * 16)) & ((1u << W) - 1u);
            *reinterpret_cast<LDSQ u32 *>(lds + LY::MASK + (i0 / (u32)W) * 256u + (u32)lane * 4u) = wsel;
            nsel += (u32)__builtin_popcount(wsel);
            return;
        }
        const u32 pbase = (u32)__builtin_amdgcn_readfirstlane((int)i0);
        guard();
        pk_unroll<W>([&](auto oc) {
            constexpr int o = decltype(oc)::value;
            this->template emit<PAR * 16 + o, o>(pbase);
        });
    }

    // state reset, warm-up over the first k-1 bases, and the words of block 0 (as FastMin::begin)
    // slot0 / step: the lane's first staging slot and its signed row stride (step 0 and slot0 = the spare slot: the lane does not stage)
    __device__ __forceinline__ void begin(u32 slot0, int step, u32 col8) {
        fl = fh_ = rl = rh_ = 0;
        bm = 0;
        nsel = 0;
        c8000 = 0x8000u;
        asm volatile("" : "+v"(c8000));  // (stays a register: as a known constant the compiler re-materialises it, or goes back to select + shift)
        tmin = 0xffffffffu;
        slot = slot0;
        sstep = step;
        if constexpr (RINGM) {
            (void)col8;
            spare = glo = gspan = 0;
            send = (u32)(LY::ROWS * LY::ROW * 8) + (u32)lane * 8u;
        } else {
            spare = (u32)(LY::PR * LY::ROW * 8) + col8;  // the column's slot in the spare row
            send = 0;
            // W staging writes from `slot` on stay in rows 0 .. PR (the spare row may be scribbled on)
            constexpr u32 RB = (u32)(LY::ROW * 8);
            glo = step < 0 ? col8 + (u32)(W - 1) * RB : col8;
            gspan = (u32)(LY::PR - W) * RB + (step < 0 ? 0u : RB);  // low lane: up to row PR - W + 1; high lane: rows W - 1 .. PR - 1
        }
        if constexpr (RINGM) load_block0_words();  // (ahead of the warm-up: whatever it loads is in flight meanwhile)
        for (int t0 = 0; t0 < k - 1; t0 += 16) {
            const u32 word = this->word((u32)t0 >> 4);
            const int nb = (k - 1 - t0) < 16 ? (k - 1 - t0) : 16;
            int j = 0;
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
            for (; j < nb; ++j) roll(*reinterpret_cast<LDSQ const u32x4 *>(lds + LY::TAB + 256 + (((word >> (2 * j)) & 3) << 4)));
        }
    }

    // nk_min: the fewest windows of a lane WITH a read (fixed-length batches: nk_max): the blocks every such lane fills need no per-lane
    // window test -- in a length-binned unit that is all but the last block or two (round 5; until then ragged batches took the test in every block)
    __device__ __forceinline__ void run(u32 nk_max, bool ok, u32 nk_min, u32 slot0, int step, u32 col8) {
        begin(slot0, step, col8);
        block<true, false, 0>(0, ok ? 1u : 0u, nk_max > (u32)W);
        u32 i0 = W;
        int last_par = 0;
        for (;;) {
            if (i0 >= nk_max) break;
            // odd block
            {
                const bool more = i0 + W < nk_max;
                if (i0 + W <= nk_min) block<false, false, 1>(i0, 1u, more);
                else block<false, true, 1>(i0, 1u, more);
            }
            i0 += W;
            last_par = 1;
            if (i0 >= nk_max) break;
            // even block
            {
                const bool more = i0 + W < nk_max;
                if (i0 + W <= nk_min) block<false, false, 0>(i0, 1u, more);
                else block<false, true, 0>(i0, 1u, more);
            }
            i0 += W;
            last_par = 0;
        }
        if (last_par) drain<1>(i0 - W);
        else drain<0>(i0 - W);
    }
};

// LDS -> HBM copy-out of one unit's staged tuples in read order (PLds staging; the arithmetic of fast_copyout, kernels_fast.hpp: owner
// of output t = the non-empty lane with the rank-th head at or before t, its table entry {A, S} puts output t at byte A + S t of the
// staged hashes).  Re-cut for latency -- the copy-out was 10 000 cycles per unit, a fifth of the kernel, for 600 instructions: three
// dependent LDS round trips per trip of four rows, 5.5 trips per unit.  Here the head words are read ONCE (lane c keeps word c, a row
// takes its word with v_readlane: no LDS), and a trip is U = 8 rows, so a unit is three trips of two round trips (table entries, then
// the staged tuples).
template <int U, class LY = PkLds>
__device__ __forceinline__ void pk_copyout(char *lds, int lane, u32 cnt, u32 excl, u32 T, u64 base, const KArgs &a) {
    constexpr int NH = LY::NHEADS;
    u64 *s_heads = reinterpret_cast<u64 *>(lds + LY::HEADS);
    u64 *s_tab64 = reinterpret_cast<u64 *>(lds + LY::CTAB);
    const u64 nzmask = __builtin_amdgcn_ballot_w64(cnt > 0);
    if (lane < NH) s_heads[lane] = 0;
    wave_sync_lds();
    if (cnt > 0) {
        constexpr int RB = LY::ROW * 8;  // bytes from one row of staged hashes to the next
        const int col8 = (lane & 31) * 8;
        const int S = lane < 32 ? RB : -RB;
        const int A = lane < 32 ? col8 - (int)excl * RB : col8 + ((int)(LY::PR - 1) + (int)excl) * RB;
        const u32 rk = __builtin_amdgcn_mbcnt_hi((u32)(nzmask >> 32), __builtin_amdgcn_mbcnt_lo((u32)nzmask, 0));
        s_tab64[rk] = ((u64)(u32)S << 32) | (u32)A;
        // bit e - 1 for a run that starts at output e >= 1 (the first run is owner 0 without a bit): the owner of output t is then the number
        // of bits below t -- a running popcount over the words before t's and one exclusive v_mbcnt in its own, nothing to shift or patch
        if (excl) atomicOr(&s_heads[(excl - 1u) >> 6], 1ULL << ((excl - 1u) & 63u));
    }
    wave_sync_lds();
    const u64 hw = lane < NH ? s_heads[lane] : 0ULL;  // word c: bit j = a lane's run starts at output 64 c + j + 1
    const u32 hw_lo = (u32)hw, hw_hi = (u32)(hw >> 32);
    u32 heads_before = 0;  // wave-uniform
    const char *sh = lds + LY::SH, *sp = lds + LY::SP;
    u64 *const gh = a.hash + base;
    u32 *const gp = a.pos + base;
    auto trip = [&](u32 t0, auto check) {
        constexpr bool CHECK = decltype(check)::value;
        u32 rank[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const u32 c = (t0 >> 6) + j;  // < 64; words from NH on read as 0: rows beyond the unit see no heads
            const u32 mlo = (u32)__builtin_amdgcn_readlane((int)hw_lo, (int)c), mhi = (u32)__builtin_amdgcn_readlane((int)hw_hi, (int)c);
            // byte offset of the owner's table entry: (bits below this lane << 3) + 8 * (bits of the words before): one v_lshl_add_u32 with
            // the scalar part as its third operand (as an initial value of v_mbcnt the scalar count costs a v_mov per row)
            rank[j] = (__builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u)) << 3) + (heads_before << 3);
            heads_before += (u32)__builtin_popcountll(((u64)mhi << 32) | mlo);
        }
        u64 ent[U];
#pragma unroll
        for (int j = 0; j < U; ++j) ent[j] = *reinterpret_cast<const u64 *>(reinterpret_cast<const char *>(s_tab64) + rank[j]);  // (< 64 entries: at most 63 bits are set)
        // (one wait for the whole trip: left alone the compiler waits before every first use -- 22 s_waitcnt per trip, each an issue slot)
        if constexpr (U == 8) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::"v"(ent[0]), "v"(ent[1]), "v"(ent[2]), "v"(ent[3]), "v"(ent[4]), "v"(ent[5]), "v"(ent[6]), "v"(ent[7]));
            __builtin_amdgcn_sched_barrier(0);
        }
        u64 hv[U];
        u32 pv[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const u32 t = t0 + 64 * j + lane;
            const int off = __mul24((int)t, (int)(u32)(ent[j] >> 32)) + (int)(u32)ent[j];  // v_mad_i32_i24
            const u32 so = (!CHECK || t < T) ? (u32)off : 0u;
            hv[j] = *reinterpret_cast<const u64 *>(sh + so);
            pv[j] = (u32)(int)*reinterpret_cast<const short *>(sp + (so >> 2));  // sign-extending read: the strand bit lands in bit 31
        }
        if constexpr (U == 8) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::"v"(hv[0]), "v"(hv[1]), "v"(hv[2]), "v"(hv[3]), "v"(hv[4]), "v"(hv[5]), "v"(hv[6]), "v"(hv[7]), "v"(pv[0]), "v"(pv[1]),
                         "v"(pv[2]), "v"(pv[3]), "v"(pv[4]), "v"(pv[5]), "v"(pv[6]), "v"(pv[7]));
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const u32 t = t0 + 64 * j + lane;
#ifdef PK_NOSTORE  // dev: the copy-out's LDS chains without its stores
            asm volatile("" ::"v"(hv[j]), "v"(pv[j]));
#else
            if (!CHECK || t < T) {
#ifndef PK_NOHASHST
                __builtin_nontemporal_store(hv[j], &gh[t]);  // write-once output: non-temporal, the tuples streaming out do not push the sequences' lines out of the L2
#else
                asm volatile("" ::"v"(hv[j]));
#endif
#ifndef PK_NOPOSST
                __builtin_nontemporal_store(pv[j] & 0x80007fffu, &gp[t]);
#else
                asm volatile("" ::"v"(pv[j]));
#endif
            }
#endif
        }
    };
    u32 t0 = 0;
    for (; t0 + 64 * U <= T; t0 += 64 * U) trip(t0, std::false_type{});
    if (t0 < T) trip(t0, std::true_type{});
    wave_sync_lds();
}

// Reads the main kernel cannot finish -- a 27-bit key tie in one of their min operations, or a staging column that filled up -- go
// to a list of READS (a.rlist: one segment per workgroup, list_append) and k_minimizer_dense<W, true>, the exact 64-bit machine with per-read slabs,
// runs them afterwards, 64 per wavefront.  History (profiles/NOTEBOOK.md): the exact machine inlined here cost the main loop 5 %
// (registers, code), as a noinline call 20 %; re-running a unit with every tuple stored straight to HBM, as k_minimizer_fast does,
// costs nine units' time (2 800 partial-line writes); a per-UNIT second pass cost a whole pass for every unit with one low-complexity
// read (2 % of the reads with a poly-A tail = 72 % of the units: 830 -> 650 Gbases/s).
#ifndef PK_TICKET
#define PK_TICKET 8u  // units per ticket
#endif
template <int W, bool LONG>
__global__ __launch_bounds__(64, 2) void k_minimizer_pk(KArgs a) {  // two waves per SIMD (the LDS staging allows eight per CU): up to 256 VGPRs
    typedef PkLds LY;
    __shared__ __attribute__((aligned(16))) char lds[LY::TOTAL];
    LDSQ char *const ldsq = (LDSQ char *)lds;
    const int lane = lane_id();
    PkTabs tabs;
    tabs.init(a.k, lane);
#ifdef PK_STAGGER  // dev: workgroups start up to PK_STAGGER x ~1 us apart
    for (u32 i = (((blockIdx.x * 2654435761u) >> 24) * (u32)PK_STAGGER) >> 8; i; --i) __builtin_amdgcn_s_sleep(32);
#endif
    const u64 slab = (u64)64 * BSK_FAST_CAP;
    const u32 col8 = (u32)(lane & 31) * 8u;
    constexpr u32 RB = (u32)(LY::ROW * 8);
    const u32 top = (u32)(LY::PR - 1) * RB + col8;  // the high lane's first slot
    // Loads and stores share ONE in-order counter on gfx9 (vmcnt), and the copy-out's store count is not a compile-time constant, so
    // any load waited for AFTER a copy-out waits for every store of that copy-out to reach memory -- 3 ms of the kernel when the
    // next unit's words were requested just ahead of the stores.  Here everything a unit reads is requested at the START of the
    // previous unit's hashing (words of unit N+1, descriptor of unit N+2) and waited for before that unit's copy-out begins, when it
    // has long arrived: after the stores nothing is waited for but LDS.
    u64 d_n1 = 0, d_cur = 0;
    PkWords pw_cur = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    bool have = false;
    const u32 lseg = a.fixcap / a.list_grid;  // this workgroup's segment of the list of reads for the exact machine (list_append)
    u32 lcur = 0;
    const u32 tku = a.tk ? a.tk : PK_TICKET;
    for (u32 unit = next_ticket(a.ticket, lane) * tku, uend = unit + tku; unit < a.nunits; ++unit, ({
             if (unit == uend) {
                 unit = next_ticket(a.ticket, lane) * tku;
                 uend = unit + tku;
             }
         })) {
        const u64 r = (u64)unit * 64 + lane;
        const bool nxt = unit + 1 != uend && unit + 1 < a.nunits;  // the next unit is this wave's too: its words and descriptor are on the way
        // Every load below is unconditional (indices beyond the batch are clamped to its last read; what such a load returns is never
        // used): a value defined on one path only would reach its wait through a copy made BEFORE the wait, i.e. from registers whose
        // load the compiler does not know to be in flight.
        const u64 rmax = a.n - 1;
        if (!have) {  // first unit of a ticket: nothing was requested ahead
            d_cur = pk_load_u64(a.desc + (r < rmax ? r : rmax));
            d_n1 = pk_load_u64(a.desc + (r + 64 < rmax ? r + 64 : rmax));
            pk_wait_loads(d_cur, d_n1);
            pw_cur = pk_load_words(a.words + (d_cur >> 24));
            pk_wait_loads(pw_cur);
        }
        PkWords pw_n1 = pk_load_words(a.words + (d_n1 >> 24));
        u64 d_n2 = pk_load_u64(a.desc + (r + 128 < rmax ? r + 128 : rmax));
        // the read's input flags (batches packed from ASCII have them): a load after the copy-out would wait for its stores
        u32 rfl = pk_load_u8(a.rflags ? a.rflags + (r < rmax ? r : rmax) : reinterpret_cast<const u8 *>(a.desc));
        const u64 d = d_cur;
        const PkWords pw = pw_cur;
        const u64 off = d >> 24, L = desc_len(a, d);
        const u64 ro = out_index(a, r, d);  // (length-binned batches: the read's own place in its chunk)
        const bool ok = r < a.n && L >= (u64)a.circ_ext && (L - (u64)a.circ_ext) + 1 >= (u64)a.k + (u64)W;
        const u32 nk = ok ? (u32)(L - a.k + 1) : 0u;
        const u32 nk_max = wave_max_u32(nk);
        const u32 nk_min = ~wave_max_u32(~(ok ? nk : 0xffffffffu));  // (over the lanes with a read)
        u32 cnt = 0, tmin_lane = 0xffffffffu;
        if (nk_max) {
            tabs.write(ldsq);  // the previous copy-out's tables took their place
            PkMin<W, LONG> pm;
            pm.w = a.words + off;
            pm.set_words(pw);
            pm.lds = ldsq;
            pm.k = a.k;
            pm.lane = lane;
            pm.nk = nk;
            // lanes l and l+32 share column l & 31, the low lane filling it from row 0 upwards, the high lane from the last row downwards;
            // a lane without a read is parked on the spare row
            const u32 spare = (u32)LY::PR * RB + col8;
            pm.run(nk_max, ok, nk_min, !ok ? spare : lane < 32 ? col8 : top, !ok ? 0 : lane < 32 ? (int)RB : -(int)RB, col8);
            if (ok) cnt = (lane < 32 ? pm.slot - col8 : top - pm.slot) / RB;
            tmin_lane = pm.tmin;
        }
        pk_wait_loads(pw_n1, d_n2, rfl);  // the next unit's words and descriptor, requested a whole hashing phase ago, are in
        d_cur = d_n1;
        pw_cur = pw_n1;
        d_n1 = d_n2;
        have = nxt;
        const u32 cnt_pair = cnt + (u32)__builtin_amdgcn_ds_bpermute((lane ^ 32) * 4, (int)cnt);
        u64 bad = __builtin_amdgcn_ballot_w64(cnt_pair >= (u32)LY::PR);  // the column's last free row takes the unselected candidates
        u64 tied = __builtin_amdgcn_ballot_w64(ok && tmin_lane < 32u);
#if defined(PK_NOSTAGE) || defined(PK_NOTIE) || defined(PK_NOTAB) || defined(PK_NOCOPY) || defined(PK_NOSTORE) || defined(PK_NOFB)
        bad = 0;
        tied = 0;
#endif
        const u64 redo = bad | tied;
        if (redo) {
            // Reads this kernel cannot finish go to a list of READS for the exact 64-bit machine (k_minimizer_dense<W, true> gathers 64 per
            // wavefront; per-read slabs, so it also takes reads that select every position): the reads with a key tie -- on real data
            // mostly low-complexity reads (poly-A / poly-G tails: every k-mer the same hash), per cent of the reads, i.e. in most units,
            // which is why the list is per read and not per unit -- and both reads of a column that filled up (0.8 % of the units on
            // random reads; next to a homopolymer tail, always).  The unit's other lanes leave normally.
            list_append(a, a.rlist, lseg, lcur, redo, lane, r);
            if ((redo >> lane) & 1) cnt = 0;  // the list pass writes this read's reference word
        }
        const u32 incl = wave_incl_scan_u32(cnt, lane);
        const u32 excl = incl - cnt;
        const u32 T = wave_bcast_u32(incl, 63);
        const u64 base = (u64)unit * slab;
#ifndef PK_NOCOPY
#ifndef PK_CU
#define PK_CU 8
#endif
        if (T) pk_copyout<PK_CU>(lds, lane, cnt, excl, T, base, a);
#endif
        if (r < a.n) {
            if (!((redo >> lane) & 1)) a.refs[ro] = ((base + excl) << 24) | cnt;  // (listed reads: the list pass writes theirs)
            u8 sbyte = ok ? BSK_ST_OK : BSK_ST_SHORT;
            if (ok && a.rflags) sbyte |= (u8)rfl;
            a.status[ro] = sbyte;
        }
    }
    list_close(a.rlist, lseg, lcur, lane);
}

#ifdef BSK_IMPL_PK
#ifndef BSK_PK_WS
#define BSK_PK_WS(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13)
#endif
// (w = 14..16 would fit the 5-bit slot numbers, but need 250..256+ VGPRs: w = 15, 16 spill -- among others registers whose asm load is
// still in flight, scripts/check_asm.py -- so they stay with k_minimizer_fast)
bool pk_minimizer_supported(int w) { return w >= 2 && w <= 13; }
u32 pk_minimizer_short_bases() { return 16u * (PKNW - 1); }  // reads up to this length never load inside the k-mer loop
int pk_minimizer_blocks_per_cu(int w) {
    int nb = 0;
    hipError_t e = hipErrorInvalidValue;
    switch (w) {
#define X(WW) \
    case WW: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_minimizer_pk<WW, false>, 64, 0); break;
        BSK_PK_WS(X)
#undef X
        default: break;
    }
    if (e != hipSuccess || nb < 1) {
        (void)hipGetLastError();
        nb = 1;
    }
    return nb;
}
void pk_minimizer_list_launch(int w, int grid, hipStream_t stream, const KArgs &a) {
    switch (w) {
#define X(WW) \
    case WW: hipLaunchKernelGGL((k_minimizer_dense<WW, true>), dim3(grid), dim3(64), 0, stream, a); break;
        BSK_PK_WS(X)
#undef X
        default: break;
    }
}
void pk_minimizer_launch(int w, bool long_reads, int grid, hipStream_t stream, const KArgs &a) {
    switch (w) {
#define X(WW)                                                                                               \
    case WW:                                                                                                \
        if (long_reads) {                                                                                   \
            hipLaunchKernelGGL((k_minimizer_pk<WW, true>), dim3(grid), dim3(64), 0, stream, a);             \
        } else {                                                                                            \
            hipLaunchKernelGGL((k_minimizer_pk<WW, false>), dim3(grid), dim3(64), 0, stream, a);            \
        }                                                                                                   \
        hipLaunchKernelGGL((k_minimizer_dense<WW, true>), dim3(grid), dim3(64), 0, stream, a);              \
        break;
        BSK_PK_WS(X)
#undef X
        default: break;
    }
}
#endif  // BSK_IMPL_PK

}  // namespace bsk
