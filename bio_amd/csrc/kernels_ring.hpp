// kernels_ring.hpp -- k_minimizer_ring<W, NQ>: the minimizer kernel for W <= 13 and 2-bit reads (NextMinimizer,
// sketches/sketch.go:205-309 -- closed form: leftmost argmin of every window, emitted when it changes).
//
// The window machine is k_minimizer_pk's (kernels_pk.hpp: packed 32-bit elements key27 | slot, selections as bits of `bm`, key ties
// detected and the read left to the exact 64-bit machine, k_minimizer_dense<W, true>).  What is new is where the tuples go:
//
//   * UNIT ROWS.  Tuple t of the read in lane l of unit u lies at  slab(u) + 64 t + l : row t of the unit's slab holds the t-th
//     tuple of each of its 64 reads (reference word bit 63, BSK_REF_ROWS: the read's tuples are 64 apart).  A row is complete as
//     soon as every lane has staged its t-th tuple -- long before the reads end -- and a complete row is 768 contiguous bytes (512 of
//     hashes, 256 of positions) that every lane writes its own piece of: no transposition through LDS, no owner look-up, no head
//     words, no ranks (k_minimizer_pk's copy-out: a seventh of the kernel), and nothing to wait for at the end of a read.
//   * RING STAGING.  LDS only aligns the lanes in time: a ring of R = 16 rows x 64 lanes x 12 bytes (hash, strand | position) --
//     12.6 KB with the two hash tables, TWELVE waves per CU where whole-read staging allowed eight.  Lane l writes entry
//     (count mod 16, l) with one ds_write2_b32 + ds_write_b32 pair and advances by the selection bit; after every block the rows below the wave's
//     smallest count leave (rg_flush).  The ring does not depend on the read length: any read the packed machine takes
//     (< 2^18 windows) runs here, and there is no column to overflow.  Lanes only conflict with themselves: entry (row, lane) is at
//     bank 3 lane + const, whatever the row -- k_minimizer_pk's staging writes spent half the LDS cycles in bank conflicts.
//   * a lane that runs more than 16 tuples ahead of the slowest lane of its unit (1.6e-5 of random 150-bp reads; homopolymer
//     tails, which are key ties anyway) or selects more than the slab's rows goes to the list of reads for the exact machine.
//   * rows that are not complete when the unit ends (lanes select 13..33 positions of a 150-bp read) are written under a lane mask:
//     12 % more bytes than the dense layout (32-byte sectors with a hole), in exchange for the above.
#pragma once
#include "kernels_pk.hpp"
#pragma clang diagnostic ignored "-Winline-asm"  // (rg_prefetch_*: the clobbered registers are reserved on purpose)

namespace bsk {

typedef u32 u32x3 __attribute__((ext_vector_type(3)));
// A ring entry is 12 bytes at a dword-aligned address.  The LDS would take a dword-aligned ds_write_b96 (unaligned-access mode of
// the HSA ABI), but it is slow in hardware: 465 Gbases/s against 1 016 for ds_write2_b32 + ds_write_b32 (measured, round 4) -- so
// the type tells the truth and the compiler splits the access (one address register, two issue slots).
#ifdef RG_B96  // dev: claim 16-byte alignment, i.e. force the single b96 access
typedef u32x3 u32x3_lds __attribute__((aligned(16)));
#else
typedef u32x3 u32x3_lds __attribute__((aligned(4)));
#endif

#ifndef BSK_RING_WAVES
#define BSK_RING_WAVES 3  // waves per SIMD the kernel is compiled for (168 VGPRs)
#endif
#ifndef BSK_RING_NQ
#define BSK_RING_NQ (BSK_RING_WAVES >= 3 ? 3 : 4)  // quads of packed words of a read kept in registers
#endif
#ifndef BSK_RING_EVERY
#define BSK_RING_EVERY 1  // blocks between two flush rounds
#endif
#ifndef BSK_RING_FIRST
#define BSK_RING_FIRST 4  // the first flush round follows this many steady blocks
#endif

#ifndef BSK_RING_XC
#define BSK_RING_XC 1  // table rows fetched per chunk, one chunk ahead (0: all W up front)
#endif
template <int W>
struct RgCfg {
    static constexpr int XC = BSK_RING_XC > 0 && BSK_RING_XC < W ? BSK_RING_XC : W;
};

struct RgLds {
    static constexpr int R = 16;        // rows of the ring (a power of two: the row is the top four bits of the lane's phase)
    static constexpr int ROWB = 768;    // a row: three planes of 64 dwords -- hash lo, hash hi, strand << 31 | position -- one dword per lane
    static constexpr int TAB = 0;       // 20 x uint4 update table
    static constexpr int TAB2 = 320;    // 16 x uint4 two-base warm-up table
    static constexpr int RING = 768;    // (a multiple of 256: the planes are reached with ds_*2st64 offsets)
    static constexpr int TOTAL = RING + R * ROWB;
    static constexpr int G = 4;         // rows leave in aligned groups of four
    // a lane this many rows ahead of the frontier makes the frontier's group leave under a lane mask.  Measured (round 4, 3e9 bases, one box):
    //   PRESS        12     13     14     15     16          a masked group costs what a full one costs (a store instruction's
    //   150 bases  1 062  1 071  1 055    959    920          price does not depend on its lane mask), so the later the better --
    //   200 bases    879    943    936    933    897          until lanes run a whole ring (16 rows) ahead of what they have sent
    //   250 bases    754    849    900    876    858          and go to the list of reads for the exact machine
    //   300 bases    643    712    767    813    806
#ifdef BSK_RING_PRESS
    static constexpr int PRESS_SHORT = BSK_RING_PRESS, PRESS_LONG = BSK_RING_PRESS;
#else
    static constexpr int PRESS_SHORT = 13, PRESS_LONG = 14;  // (reads up to / beyond ring_minimizer_short_bases())
#endif
#ifndef BSK_RING_LAZY
#define BSK_RING_LAZY 6
#endif
    static constexpr int LAZY = BSK_RING_LAZY;  // a lane behind the frontier catches up once it has this many tuples waiting
    static_assert(TOTAL <= 13312, "twelve waves per CU");
};

// NQ x 4 packed words of a read in registers, loaded one unit ahead by loads the s_waitcnt pass does not see (kernels_pk.hpp)
template <int NQ>
struct RgWords {
    u32x4 q[NQ];
};
template <int NQ>
__device__ __forceinline__ RgWords<NQ> rg_load_words(const u32 *p) {
    RgWords<NQ> r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(r.q[0]) : "v"(p));
    if constexpr (NQ > 1) asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=&v"(r.q[1]) : "v"(p));
    if constexpr (NQ > 2) asm volatile("global_load_dwordx4 %0, %1, off offset:32" : "=&v"(r.q[2]) : "v"(p));
    if constexpr (NQ > 3) asm volatile("global_load_dwordx4 %0, %1, off offset:48" : "=&v"(r.q[3]) : "v"(p));
    if constexpr (NQ > 4) asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=&v"(r.q[4]) : "v"(p));
    static_assert(NQ <= 5, "RgWords");
    return r;
}
template <int NQ>
__device__ __forceinline__ void rg_wait_loads(RgWords<NQ> &p) {
    if constexpr (NQ == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(p.q[0]), "+v"(p.q[1]), "+v"(p.q[2])::"memory");
    else if constexpr (NQ == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(p.q[0]), "+v"(p.q[1]), "+v"(p.q[2]), "+v"(p.q[3])::"memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(p.q[0]), "+v"(p.q[1]), "+v"(p.q[2]), "+v"(p.q[3]), "+v"(p.q[4])::"memory");
}
template <int NQ>
__device__ __forceinline__ void rg_wait_loads(RgWords<NQ> &p, u64 &d0, u32 &f) {
    if constexpr (NQ == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(p.q[0]), "+v"(p.q[1]), "+v"(p.q[2]), "+v"(d0), "+v"(f)::"memory");
    else if constexpr (NQ == 4)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(p.q[0]), "+v"(p.q[1]), "+v"(p.q[2]), "+v"(p.q[3]), "+v"(d0), "+v"(f)::"memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(p.q[0]), "+v"(p.q[1]), "+v"(p.q[2]), "+v"(p.q[3]), "+v"(p.q[4]), "+v"(d0), "+v"(f)::"memory");
}

// Three waves per SIMD: the loads of the next unit land in the TOP registers of the 168 (v152 .. v166; tuples start on even registers), which the kernel keeps
// out of the allocator's hands (amdgpu_num_vgpr below; the clobber lists make the kernel descriptor count them), and are moved into
// ordinary registers behind the wait.  scripts/check_asm.py checks that nothing outside these two blocks names a register up there.
#if BSK_RING_WAVES >= 3
#if BSK_RING_NQ == 3
#define RG_FREE_VGPRS 144  // v144..v151: RgMin::issue_next, v152..v166: rg_prefetch_issue
__device__ __forceinline__ void rg_prefetch_issue(const u32 *pw, const u64 *pd, const u8 *pf) {
    asm volatile("global_load_dwordx4 v[152:155], %0, off\n\tglobal_load_dwordx4 v[156:159], %0, off offset:16\n\t"
                 "global_load_dwordx4 v[160:163], %0, off offset:32\n\tglobal_load_dwordx2 v[164:165], %1, off\n\tglobal_load_ubyte v166, %2, off"
                 :
                 : "v"(pw), "v"(pd), "v"(pf)
                 : "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166");
}
__device__ __forceinline__ void rg_prefetch_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// (the values stay up there until the unit ends: no allocatable register holds them during the block loop)
__device__ __forceinline__ void rg_prefetch_take(RgWords<3> &p, u64 &d, u32 &f) {
    u32 dl, dh;
    asm volatile("v_mov_b32 %0, v152\n\tv_mov_b32 %1, v153\n\tv_mov_b32 %2, v154\n\tv_mov_b32 %3, v155\n\t"
                 "v_mov_b32 %4, v156\n\tv_mov_b32 %5, v157\n\tv_mov_b32 %6, v158\n\tv_mov_b32 %7, v159\n\t"
                 "v_mov_b32 %8, v160\n\tv_mov_b32 %9, v161\n\tv_mov_b32 %10, v162\n\tv_mov_b32 %11, v163\n\t"
                 "v_mov_b32 %12, v164\n\tv_mov_b32 %13, v165\n\tv_mov_b32 %14, v166"
                 : "=&v"(p.q[0].x), "=&v"(p.q[0].y), "=&v"(p.q[0].z), "=&v"(p.q[0].w), "=&v"(p.q[1].x), "=&v"(p.q[1].y), "=&v"(p.q[1].z), "=&v"(p.q[1].w),
                   "=&v"(p.q[2].x), "=&v"(p.q[2].y), "=&v"(p.q[2].z), "=&v"(p.q[2].w), "=&v"(dl), "=&v"(dh), "=&v"(f)
                 :
                 : "memory", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166");
    d = ((u64)dh << 32) | dl;
}
#else
#error "three waves per SIMD: BSK_RING_NQ must be 3"
#endif
#define RG_KERNEL_ATTR __attribute__((amdgpu_num_vgpr(RG_FREE_VGPRS / 2)))  // (gfx90a+ doubles the request: VGPRs + AGPRs of the unified file)
#else
#define RG_KERNEL_ATTR
#endif

template <int N>
struct RgVec;
template <>
struct RgVec<12> {
    typedef u32 type __attribute__((ext_vector_type(12)));
};
template <>
struct RgVec<16> {
    typedef u32 type __attribute__((ext_vector_type(16)));
};
template <>
struct RgVec<20> {
    typedef u32 type __attribute__((ext_vector_type(32)));  // (no 20-register class: the tuple is 32 wide)
};

// Rows tg .. tg + 3 of the unit (tg a multiple of four: the four ring rows do not wrap) leave the ring: every lane sends its own
// entries to its column of the unit's slab.  va = the lane's byte offset in ring row tg & 15, voh / vop = its byte offsets in the
// slab's hash / position rows tg, sh / sp = the slab (wave-uniform).
// MODE 0: every lane has all four rows (a complete group of a wave whose lanes all hold a read).
// MODE 1: row j of a lane leaves iff j < hi (signed).   MODE 2: iff lo <= j < hi (a lane that fell behind the frontier catches up).
template <int MODE>
__device__ __forceinline__ void rg_group(u32 va, u32 voh, u32 vop, const u64 *sh, const u32 *sp, int hi, int lo) {
    u64 h0, h1, h2, h3, sv;
    u32 p0, p1, p2, p3;
#define RG_READS                                                                                                           \
    "ds_read2st64_b32 %[h0], %[va] offset0:3 offset1:4\n\tds_read_b32 %[p0], %[va] offset:1280\n\t"                        \
    "ds_read2st64_b32 %[h1], %[va] offset0:6 offset1:7\n\tds_read_b32 %[p1], %[va] offset:2048\n\t"                        \
    "ds_read2st64_b32 %[h2], %[va] offset0:9 offset1:10\n\tds_read_b32 %[p2], %[va] offset:2816\n\t"                       \
    "ds_read2st64_b32 %[h3], %[va] offset0:12 offset1:13\n\tds_read_b32 %[p3], %[va] offset:3584\n\t"
#ifdef RG_NOSTORE  // dev: the flush's LDS reads without its stores
#define RG_ST(j, o8, o4) ""
#elif defined(RG_HASHONLY)
#define RG_ST(j, o8, o4) "global_store_dwordx2 %[voh], %[h" #j "], %[sh] offset:" #o8 " nt\n\t"
#elif defined(RG_POSONLY)
#define RG_ST(j, o8, o4) "global_store_dword %[vop], %[p" #j "], %[sp] offset:" #o4 " nt\n\t"
#elif defined(RG_PLAIN)
#define RG_ST(j, o8, o4) "global_store_dwordx2 %[voh], %[h" #j "], %[sh] offset:" #o8 "\n\tglobal_store_dword %[vop], %[p" #j "], %[sp] offset:" #o4 "\n\t"
#else
#define RG_ST(j, o8, o4) "v_alignbit_b32 %[p" #j "], %[p" #j "], %[p" #j "], 1\n\tglobal_store_dwordx2 %[voh], %[h" #j "], %[sh] offset:" #o8 " nt\n\tglobal_store_dword %[vop], %[p" #j "], %[sp] offset:" #o4 " nt\n\t"
#endif
#define RG_OUT [h0] "=&v"(h0), [h1] "=&v"(h1), [h2] "=&v"(h2), [h3] "=&v"(h3), [p0] "=&v"(p0), [p1] "=&v"(p1), [p2] "=&v"(p2), [p3] "=&v"(p3)
#ifdef RG_X4TEST  // dev (timing only, results wrong): a group leaves as three 16-byte stores per lane
    {
        u32x4 H01, H23, PP;
        const u32 voh2 = voh + (voh & 511u) * 3u, vop2 = vop + (vop & 255u) * 3u;  // 32 lane + 2048 (tg / 4), 16 lane + 1024 (tg / 4)
        if constexpr (MODE == 0) {
            asm volatile("ds_read_b128 %[a], %[va] offset:768\n\tds_read_b128 %[b], %[va] offset:784\n\tds_read_b128 %[c], %[va] offset:800\n\t"
                         "s_waitcnt lgkmcnt(0)\n\tglobal_store_dwordx4 %[voh], %[a], %[sh] nt\n\tglobal_store_dwordx4 %[voh], %[b], %[sh] offset:16 nt\n\t"
                         "global_store_dwordx4 %[vop], %[c], %[sp] nt"
                         : [a] "=&v"(H01), [b] "=&v"(H23), [c] "=&v"(PP)
                         : [va] "v"(va & ~15u), [voh] "v"(voh2), [vop] "v"(vop2), [sh] "s"(sh), [sp] "s"(sp)
                         : "memory");
        } else {
            asm volatile("ds_read_b128 %[a], %[va] offset:768\n\tds_read_b128 %[b], %[va] offset:784\n\tds_read_b128 %[c], %[va] offset:800\n\t"
                         "s_mov_b64 %[sv], exec\n\tv_cmpx_lt_i32 0, %[hi]\n\t"
                         "s_waitcnt lgkmcnt(0)\n\tglobal_store_dwordx4 %[voh], %[a], %[sh] nt\n\tglobal_store_dwordx4 %[voh], %[b], %[sh] offset:16 nt\n\t"
                         "global_store_dwordx4 %[vop], %[c], %[sp] nt\n\ts_mov_b64 exec, %[sv]"
                         : [a] "=&v"(H01), [b] "=&v"(H23), [c] "=&v"(PP), [sv] "=&s"(sv)
                         : [va] "v"(va & ~15u), [voh] "v"(voh2), [vop] "v"(vop2), [sh] "s"(sh), [sp] "s"(sp), [hi] "v"(hi - lo * (MODE == 2 ? 1 : 0))
                         : "memory", "vcc");
        }
        return;
    }
#endif
    if constexpr (MODE == 0) {
        asm volatile(RG_READS "s_waitcnt lgkmcnt(4)\n\t" RG_ST(0, 0, 0) RG_ST(1, 512, 256) "s_waitcnt lgkmcnt(0)\n\t" RG_ST(2, 1024, 512) RG_ST(3, 1536, 768)
                     : RG_OUT
                     : [va] "v"(va), [voh] "v"(voh), [vop] "v"(vop), [sh] "s"(sh), [sp] "s"(sp)
                     : "memory");
    } else if constexpr (MODE == 1) {  // the masks only shrink from row to row
        asm volatile(RG_READS "s_mov_b64 %[sv], exec\n\t"
                     "v_cmpx_lt_i32 0, %[hi]\n\ts_waitcnt lgkmcnt(6)\n\t" RG_ST(0, 0, 0)
                     "v_cmpx_lt_i32 1, %[hi]\n\ts_waitcnt lgkmcnt(4)\n\t" RG_ST(1, 512, 256)
                     "v_cmpx_lt_i32 2, %[hi]\n\ts_waitcnt lgkmcnt(2)\n\t" RG_ST(2, 1024, 512)
                     "v_cmpx_lt_i32 3, %[hi]\n\ts_waitcnt lgkmcnt(0)\n\t" RG_ST(3, 1536, 768)
                     "s_mov_b64 exec, %[sv]"
                     : RG_OUT, [sv] "=&s"(sv)
                     : [va] "v"(va), [voh] "v"(voh), [vop] "v"(vop), [sh] "s"(sh), [sp] "s"(sp), [hi] "v"(hi)
                     : "memory", "vcc");
    } else {
        asm volatile(RG_READS "s_mov_b64 %[sv], exec\n\t"
                     "v_cmpx_lt_i32 0, %[hi]\n\tv_cmpx_ge_i32 0, %[lo]\n\ts_waitcnt lgkmcnt(6)\n\t" RG_ST(0, 0, 0) "s_mov_b64 exec, %[sv]\n\t"
                     "v_cmpx_lt_i32 1, %[hi]\n\tv_cmpx_ge_i32 1, %[lo]\n\ts_waitcnt lgkmcnt(4)\n\t" RG_ST(1, 512, 256) "s_mov_b64 exec, %[sv]\n\t"
                     "v_cmpx_lt_i32 2, %[hi]\n\tv_cmpx_ge_i32 2, %[lo]\n\ts_waitcnt lgkmcnt(2)\n\t" RG_ST(2, 1024, 512) "s_mov_b64 exec, %[sv]\n\t"
                     "v_cmpx_lt_i32 3, %[hi]\n\tv_cmpx_ge_i32 3, %[lo]\n\ts_waitcnt lgkmcnt(0)\n\t" RG_ST(3, 1536, 768) "s_mov_b64 exec, %[sv]"
                     : RG_OUT, [sv] "=&s"(sv)
                     : [va] "v"(va), [voh] "v"(voh), [vop] "v"(vop), [sh] "s"(sh), [sp] "s"(sp), [hi] "v"(hi), [lo] "v"(lo)
                     : "memory", "vcc");
    }
#undef RG_READS
#undef RG_ST
#undef RG_OUT
}

__device__ __forceinline__ u32 wave_min_u32(u32 v) {  // DPP ladder with min (lanes without a source keep their own value)
#define BSK_DPPM(v, ctrl, rmask) ((u32)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)(v), (ctrl), (rmask), 0xf, false))
    u32 t;
    t = BSK_DPPM(v, 0x111, 0xf); v = t < v ? t : v;
    t = BSK_DPPM(v, 0x112, 0xf); v = t < v ? t : v;
    t = BSK_DPPM(v, 0x114, 0xf); v = t < v ? t : v;
    t = BSK_DPPM(v, 0x118, 0xf); v = t < v ? t : v;
    t = BSK_DPPM(v, 0x142, 0xa); v = t < v ? t : v;
    t = BSK_DPPM(v, 0x143, 0xc); v = t < v ? t : v;
#undef BSK_DPPM
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
}

// the packed window machine + ring staging of one read per lane
// NQ:   the read's first 4 NQ packed words live in registers
// LONG: reads of more than 16 (4 NQ - 1) bases (their further words are loaded inside the k-mer loop)
template <int W, int NQ, bool LONG>
struct RgMin {
    typedef RgLds LY;
    static constexpr int NW = 4 * NQ;
    const u32 *__restrict__ w;
    LDSQ char *lds;
    int k, lane;
    u32 nk;
    u32 fl, fh_, rl, rh_;
    u32 S[W];    // packed suffix minima of the previous block (then raw packed values of the current one)
    u32 HL[W], HH[W];  // canonical hashes of the previous block's slots, slot by slot replaced by the current block's
    lmask RV[W];       // and the lanes whose reverse strand won there: the strand bit enters the staged position word as a carry
                       // (v_addc_co_u32: one instruction where a select at the k-mer and an add at the staging step were two)
    u32 P, bm, tmin;
    u32 phase;   // (staged tuples mod 16) << 28 | ceil(lane * 2^22 / 3): mul_hi(phase, 16 * 768) = 768 row + 4 lane, the byte offset of the lane's next ring entry
    u32 c;       // tuples staged so far
    typename RgVec<NW>::type wr;

    __device__ __forceinline__ void set_words(const RgWords<NQ> &p) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            wr[4 * j] = p.q[j].x;
            wr[4 * j + 1] = p.q[j].y;
            wr[4 * j + 2] = p.q[j].z;
            wr[4 * j + 3] = p.q[j].w;
        }
    }
    // LONG: the words a block needs beyond the registers' are requested TWO blocks ahead, right behind a round's stores, and waited
    // for before the next round's stores (vmcnt is one in-order counter: a load waited for behind a round waits for its stores).
    // They land in reserved registers (rg_prefetch_issue: v144..v147 for the odd blocks, v148..v151 for the even ones) -- values
    // the compiler knows of were copied while their loads were in flight.
    template <int PAR>
    __device__ __forceinline__ void issue_next(u32 i0n) {  // the words of the block that starts at k-mer i0n (parity PAR)
        if constexpr (LONG) {
            const u32 t0 = i0n + (u32)k - 1, p0 = i0n - 1u;
            if (PAR) asm volatile("global_load_dwordx2 v[144:145], %0, off\n\tglobal_load_dwordx2 v[146:147], %1, off" ::"v"(w + (t0 >> 4)), "v"(w + (p0 >> 4)) : "v144", "v145", "v146", "v147");
            else asm volatile("global_load_dwordx2 v[148:149], %0, off\n\tglobal_load_dwordx2 v[150:151], %1, off" ::"v"(w + (t0 >> 4)), "v"(w + (p0 >> 4)) : "v148", "v149", "v150", "v151");
        }
    }
    __device__ __forceinline__ void wait_next() {
        if constexpr (LONG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    template <int PAR>
    __device__ __forceinline__ void word2(u32 i, bool in, u32 &lo, u32 &hi) const {
        const u32 iu = (u32)__builtin_amdgcn_readfirstlane((int)i);
        if (!LONG || iu + 1 < (u32)NW) {
            lo = wr[iu], hi = wr[iu + 1];
        } else if (PAR) {
            if (in) asm volatile("v_mov_b32 %0, v144\n\tv_mov_b32 %1, v145" : "=v"(lo), "=v"(hi));
            else asm volatile("v_mov_b32 %0, v146\n\tv_mov_b32 %1, v147" : "=v"(lo), "=v"(hi));
        } else {
            if (in) asm volatile("v_mov_b32 %0, v148\n\tv_mov_b32 %1, v149" : "=v"(lo), "=v"(hi));
            else asm volatile("v_mov_b32 %0, v150\n\tv_mov_b32 %1, v151" : "=v"(lo), "=v"(hi));
        }
    }
    __device__ __forceinline__ u32 word(u32 i) const {
        const u32 iu = (u32)__builtin_amdgcn_readfirstlane((int)i);
        if (!LONG || iu < (u32)NW) return wr[iu];
        return w[iu];
    }
    __device__ __forceinline__ void roll(u32x4 x) {
        const u32 a = __builtin_amdgcn_alignbit(fl, fh_, 31), b = __builtin_amdgcn_alignbit(fh_, fl, 31);
        const u32 c_ = __builtin_amdgcn_alignbit(rh_, rl, 1), d = __builtin_amdgcn_alignbit(rl, rh_, 1);
        fl = a ^ x.x;
        fh_ = b ^ x.y;
        rl = c_ ^ x.z;
        rh_ = d ^ x.w;
    }
    __device__ __forceinline__ void roll2(u32x4 x) {
        const u32 a = __builtin_amdgcn_alignbit(fl, fh_, 30), b = __builtin_amdgcn_alignbit(fh_, fl, 30);
        const u32 c_ = __builtin_amdgcn_alignbit(rh_, rl, 2), d = __builtin_amdgcn_alignbit(rl, rh_, 2);
        fl = a ^ x.x;
        fh_ = b ^ x.y;
        rl = c_ ^ x.z;
        rh_ = d ^ x.w;
    }

    // one staging step: slot o of the block whose slot 0 is k-mer pbase (wave-uniform).  The entry is written whether the slot was
    // selected or not; the ring position advances by the selection bit.
    template <int IDX, int O>
    __device__ __forceinline__ void emit(u32 pbase2) {
        u32 b;
        asm("v_bfe_u32 %0, %1, %2, 1" : "=v"(b) : "v"(bm), "n"(IDX));
        u32 pv;  // 2 * position + strand (rg_group turns it round by one bit): 2 pbase in a register, 2 O inline, the strand as the carry in
        asm("v_addc_co_u32 %0, vcc, %1, %2, %3" : "=v"(pv) : "v"(pbase2), "n"(2 * O), "s"(RV[O]) : "vcc");
        const u32 at = __umulhi(phase, (u32)(LY::R * LY::ROWB));
#if defined(RG_MASKST)  // dev: the staging writes under the selection bit's lane mask, no branch
        {
            const u32 a0 = (u32)(uintptr_t)(lds + LY::RING) + at;
            u64 sv;
            asm volatile("v_cmp_ne_u32_e32 vcc, 0, %1\n\ts_and_saveexec_b64 %0, vcc\n\tds_write2st64_b32 %2, %3, %4 offset1:1\n\tds_write_b32 %2, %5 offset:512\n\ts_mov_b64 exec, %0"
                         : "=&s"(sv) : "v"(b), "v"(a0), "v"(HL[O]), "v"(HH[O]), "v"(pv) : "vcc", "memory");
        }
#elif !defined(RG_NOSTAGE)
        LDSQ u32 *const e = reinterpret_cast<LDSQ u32 *>(lds + LY::RING + at);  // the three planes of the row: one ds_write2st64_b32 + one ds_write_b32
        e[0] = HL[O];
        e[64] = HH[O];
        e[128] = pv;
#else
        asm volatile("" ::"v"(HL[O]), "v"(HH[O]), "v"(pv), "v"(at));
#endif
        asm("v_lshl_add_u32 %0, %1, 28, %0" : "+v"(phase) : "v"(b));  // the row wraps with the register (as C the compiler shifts, masks and adds)
    }

    // FIRST: block 0 (nothing leaves at slot 0, no window is complete before its last step, nothing to emit; okbit = 0 for lanes
    //        without a read)
    // RAG:   windows end per lane (ragged batch, or the wave's last, partial block)
    // PAR:   parity of the block: its slots are idx PAR*16 + o, the previous block's (1-PAR)*16 + o
    template <bool FIRST, bool RAG, int PAR>
    __device__ __forceinline__ void block(u32 i0, u32 okbit, bool suffix) {
        constexpr int CB = PAR * 16, PB = (1 - PAR) * 16;
        constexpr int XC = RgCfg<W>::XC;
        const u32 t0 = i0 + (u32)k - 1, p0 = FIRST ? 0u : i0 - 1u;
        u32 in_lo, in_hi, out_lo, out_hi;
        this->template word2<PAR>(t0 >> 4, true, in_lo, in_hi);
        this->template word2<PAR>(p0 >> 4, false, out_lo, out_hi);
        const u32 cinb = __builtin_amdgcn_alignbit(in_hi, in_lo, (t0 & 15) * 2);  // code of slot o at bits [2o, 2o+2)
        u32 coutb;
        if (FIRST) coutb = out_lo << 2;  // slot 0: nothing leaves; slot o >= 1 sees base o-1
        else coutb = __builtin_amdgcn_alignbit(out_hi, out_lo, (p0 & 15) * 2);
        // table offsets: nibble j of E / O = (out << 2 | in) of slot 2j / 2j+1, so a slot's row offset is (word >> n) & 0xF0
        const u32 E = (cinb & 0x33333333u) | ((coutb & 0x33333333u) << 2);
        const u32 O = ((cinb >> 2) & 0x33333333u) | (coutb & 0xCCCCCCCCu);
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
                xs[o] = *reinterpret_cast<LDSQ const u32x4 *>(lds + LY::TAB + a);
            }
        };
        fetch(0);
        if (XC < W) fetch(XC);
        u32 pbase = 2u * (i0 - (u32)W);  // (twice the first k-mer of the previous block, in a VGPR: the carry form takes one scalar operand, the lane mask)
        asm volatile("" : "+v"(pbase));
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
            HL[o] = hl;
            HH[o] = hh;
            RV[o] = rev;
            u32 pk;  // (hh & ~31) | idx
            asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(pk) : "v"(hh), "s"(0xffffffe0u), "n"(CB + o));
            if (o == 0) {
                P = pk;
            } else {
#ifndef RG_NOTIE
                const u32 d = pk ^ P;
                tmin = tmin < d ? tmin : d;
#endif
                P = P < pk ? P : pk;
            }
            if (!FIRST || o == W - 1) {
                u32 m = P;
                if constexpr (o != W - 1) {
#ifndef RG_NOTIE
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
#ifndef RG_NOTIE
                const u32 d = S[q] ^ S[q + 1];
                tmin = tmin < d ? tmin : d;
#endif
                S[q] = S[q] < S[q + 1] ? S[q] : S[q + 1];
            }
        }
        if (!FIRST) {  // the previous block's slots are all staged
            c += (u32)__builtin_popcount(bm & (PAR ? 0x0000ffffu : 0xffff0000u));
            bm &= PAR ? 0xffff0000u : 0x0000ffffu;
        }
    }

    // the last block's own slots
    template <int PAR>
    __device__ __forceinline__ void drain(u32 i0) {
        u32 pbase = 2u * i0;
        asm volatile("" : "+v"(pbase));
        pk_unroll<W>([&](auto oc) {
            constexpr int o = decltype(oc)::value;
            this->template emit<PAR * 16 + o, o>(pbase);
        });
        c += (u32)__builtin_popcount(bm);
    }

    // state reset and warm-up over the first k-1 bases (as PkMin::begin)
    __device__ __forceinline__ void begin() {
        fl = fh_ = rl = rh_ = 0;
        bm = 0;
        c = 0;
        tmin = 0xffffffffu;
        phase = ((u32)lane * (1u << 22) + 2u) / 3u;
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
};

#ifndef RG_TICKET
#define RG_TICKET 8u  // units per ticket
#endif
template <int W, int NQ, bool LONG>
__global__ __launch_bounds__(64, BSK_RING_WAVES) RG_KERNEL_ATTR void k_minimizer_ring(KArgs a) {
    typedef RgLds LY;
    __shared__ __attribute__((aligned(16))) char lds[LY::TOTAL];
    LDSQ char *const ldsq = (LDSQ char *)lds;
    const int lane = lane_id();
    {
        PkTabs tabs;  // (same rows, same offsets: TAB = 0, TAB2 = 320)
        tabs.init(a.k, lane);
        tabs.write(ldsq);
    }
#ifdef RG_STAGGER  // dev: workgroups start up to RG_STAGGER x ~1 us apart (are the waves' store bursts in phase?)
    for (u32 i = (((blockIdx.x * 2654435761u) >> 24) * (u32)RG_STAGGER) >> 8; i; --i) __builtin_amdgcn_s_sleep(32);
#endif
    const u32 rows = a.unit_rows;  // rows of a unit's slab
    const u64 slab = (u64)64 * rows;
    const bool uniform_batch = a.uniform_len != 0;
    const u32 lane4 = (u32)lane * 4u, lane8 = (u32)lane * 8u;
    u64 d_n1 = 0, d_cur = 0;
    RgWords<NQ> pw_cur;
#pragma unroll
    for (int j = 0; j < NQ; ++j) pw_cur.q[j] = (u32x4){0, 0, 0, 0};
    bool have = false;
    const u32 lseg = a.fixcap / a.list_grid;  // this workgroup's segment of the list of reads for the exact machine (list_append)
    u32 lcur = 0;
    const u32 tku = a.tk ? a.tk : RG_TICKET;
    for (u32 unit = next_ticket(a.ticket, lane) * tku, uend = unit + tku; unit < a.nunits; ++unit, ({
             if (unit == uend) {
                 unit = next_ticket(a.ticket, lane) * tku;
                 uend = unit + tku;
             }
         })) {
        const u64 r = (u64)unit * 64 + lane;
        const bool nxt = unit + 1 != uend && unit + 1 < a.nunits;  // the next unit is this wave's too: its words and descriptor are on the way
        // gfx9 counts loads and stores in ONE in-order vmcnt, and this kernel stores after every block: what a unit reads is requested
        // at the START of the previous unit (words of unit N+1, descriptor of unit N+2) by loads the compiler's s_waitcnt pass does not
        // see (kernels_pk.hpp), and waited for after that unit's third block, before its first row leaves -- behind that point a wait
        // would also wait for stores.  Every such load is unconditional (indices beyond the batch are clamped to its last read).
        const u64 rmax = a.n - 1;
        if (!have) {  // first unit of a ticket: nothing was requested ahead
            d_cur = pk_load_u64(a.desc + (r < rmax ? r : rmax));
            d_n1 = pk_load_u64(a.desc + (r + 64 < rmax ? r + 64 : rmax));
            pk_wait_loads(d_cur, d_n1);
            pw_cur = rg_load_words<NQ>(a.words + (d_cur >> 24));
            rg_wait_loads<NQ>(pw_cur);
        }
#if BSK_RING_WAVES >= 3
        // (under the 168-VGPR cap the register allocator spills exactly these values -- long-lived, unused in the block loop -- while
        // their loads are in flight; so the loads land in registers it does not allocate at all: rg_prefetch_issue / _take)
        rg_prefetch_issue(a.words + (d_n1 >> 24), a.desc + (r + 128 < rmax ? r + 128 : rmax),
                          a.rflags ? a.rflags + (r < rmax ? r : rmax) : reinterpret_cast<const u8 *>(a.desc));
#else
        RgWords<NQ> pw_n1 = rg_load_words<NQ>(a.words + (d_n1 >> 24));
        u64 d_n2 = pk_load_u64(a.desc + (r + 128 < rmax ? r + 128 : rmax));
        u32 rfl = pk_load_u8(a.rflags ? a.rflags + (r < rmax ? r : rmax) : reinterpret_cast<const u8 *>(a.desc));
#endif
        const u64 d = d_cur;
        const u64 off = d >> 24, L = desc_len(a, d);
        const u64 ro = out_index(a, r, d);  // (length-binned batches: the read's own place in its chunk)
        const bool ok = r < a.n && L >= (u64)a.circ_ext && (L - (u64)a.circ_ext) + 1 >= (u64)a.k + (u64)W;
        const u32 nk = ok ? (u32)(L - a.k + 1) : 0u;
        const u32 nk_max = wave_max_u32(nk);
        const bool uniform = uniform_batch && __builtin_amdgcn_ballot_w64(ok) == ~0ULL;  // (a lane without a read must not select)
#if defined(RG_L2STORE) || defined(RG_SMALLSLAB)  // dev: every unit of a workgroup writes the same slab (results wrong)
        const u64 base = (u64)blockIdx.x * slab;
#else
        const u64 base = (u64)unit * slab;
#endif
        const u64 *const sh = a.hash + base;  // the unit's slab (wave-uniform)
        const u32 *const sp = a.pos + base;
        RgMin<W, NQ, LONG> pm;
        pm.w = a.words + off;
        pm.set_words(pw_cur);
        pm.lds = ldsq;
        pm.k = a.k;
        pm.lane = lane;
        pm.nk = nk;
        // Rows leave in aligned groups of four.  T (wave-uniform, a multiple of four) is the frontier: group T leaves when every lane
        // that is still selecting has staged its tuple T + 3 (a complete group: whole lines, no mask in a wave without gaps), or when a lane
        // is PRESS rows ahead of T and would soon write over its own waiting entries (then under a lane mask: the lanes behind send their
        // pieces of the group when they get there -- "catch up" below).  g = the lane's tuples that have left; after a round
        // g = min(c, T).  A lane that gets a whole ring ahead of its g all the same has lost entries and goes to the list (`lost`).
        u32 T = 0, g = 0;
        lmask lost = 0;
        auto group = [&](u32 tg, auto mode, int hi, int lo) {
            const u32 va = lane4 + (tg & (u32)(LY::R - 1)) * (u32)LY::ROWB;
#ifdef RG_SMALLSLAB  // dev: every group of a wave lands on the same two groups of rows (the stores stay in the L2; results wrong)
            rg_group<decltype(mode)::value>(va, lane8 + (tg & 4u) * 512u, lane4 + (tg & 4u) * 256u, sh, sp, hi, lo);
#else
            rg_group<decltype(mode)::value>(va, lane8 + tg * 512u, lane4 + tg * 256u, sh, sp, hi, lo);
#endif
        };
        // catch up: lanes behind the frontier send the tuples (below `top` <= T) they have staged since they last sent any --
        // during the read only once some lane has `lazy` tuples waiting (a group sent for one tuple costs what a full one costs)
        auto catch_up = [&](u32 top, u32 lazy) {
            if (!__builtin_amdgcn_ballot_w64(top - g >= lazy)) return;
            const lmask need = __builtin_amdgcn_ballot_w64(g < top);
            for (u32 tg = T - (u32)LY::G;; tg -= (u32)LY::G) {
                const lmask here = need & __builtin_amdgcn_ballot_w64(g < tg + (u32)LY::G);  // lanes with rows in this group or below it
                if (here & __builtin_amdgcn_ballot_w64(top > tg)) group(tg, std::integral_constant<int, 2>{}, (int)(top - tg), (int)(g - tg));
                if (!(here & __builtin_amdgcn_ballot_w64(g < tg)) || tg == 0) break;
            }
            g = top;
        };
        auto round = [&](u32 i0_done) {
            lost |= __builtin_amdgcn_ballot_w64(pm.c - g >= (u32)LY::R);
            catch_up(pm.c < T ? pm.c : T, (u32)LY::LAZY);
            // a lane that still has tuples waiting below the frontier (fewer than LAZY) sends nothing above it either: what waits stays one run
            const bool lagging = g < T;
            const lmask lag = __builtin_amdgcn_ballot_w64(lagging);
            const lmask act = uniform ? ~0ULL : __builtin_amdgcn_ballot_w64(pm.nk > i0_done);  // (a lane whose windows have ended holds no group back)
            for (;;) {
                if (T + (u32)LY::G > rows) break;
                const lmask behind = act & __builtin_amdgcn_ballot_w64(pm.c < T + (u32)LY::G);
                if (behind && !__builtin_amdgcn_ballot_w64(pm.c >= T + (u32)(LONG ? LY::PRESS_LONG : LY::PRESS_SHORT))) break;
                if (!behind && !lag && uniform) group(T, std::integral_constant<int, 0>{}, 0, 0);
                else group(T, std::integral_constant<int, 1>{}, lagging ? 0 : (int)(pm.c - T), 0);
                T += (u32)LY::G;
            }
            g = lagging ? g : (pm.c < T ? pm.c : T);
        };
        // Block 0, then the one wait for the next unit's words (requested before the warm-up), then blocks and flush rounds in turn.
        pm.begin();
        pm.template block<true, false, 0>(0, ok ? 1u : 0u, nk_max > (u32)W);
#if BSK_RING_WAVES < 3
        rg_wait_loads<NQ>(pw_n1, d_n2, rfl);
#endif
        u32 i0 = W;
        int last_par = 0;
        u32 nround = 0;
        pm.template issue_next<1>(W);  // (block 1 starts right away: LONG reads' first blocks read the registers' words anyway)
        pm.template issue_next<0>(2 * W);
        for (;;) {
            if (i0 >= nk_max) break;
            {
                const bool more = i0 + W < nk_max;
                if (uniform && i0 + W <= nk_max) pm.template block<false, false, 1>(i0, 1u, more);
                else pm.template block<false, true, 1>(i0, 1u, more);
            }
            i0 += W;
            last_par = 1;
            if (i0 >= nk_max) break;
            pm.wait_next();  // the words of the block that starts now (requested a block ago), before this round's stores
            ++nround;
#if BSK_RING_WAVES >= 3
            if (nround == BSK_RING_FIRST) rg_prefetch_wait();  // (a bare s_waitcnt: the values stay in their reserved registers)
#endif
            if (nround >= BSK_RING_FIRST) round(i0 - W);
            pm.template issue_next<1>(i0 + W);  // for the block after the next
            {
                const bool more = i0 + W < nk_max;
                if (uniform && i0 + W <= nk_max) pm.template block<false, false, 0>(i0, 1u, more);
                else pm.template block<false, true, 0>(i0, 1u, more);
            }
            i0 += W;
            last_par = 0;
            if (i0 >= nk_max) break;
            pm.wait_next();
            ++nround;
#if BSK_RING_WAVES >= 3
            if (nround == BSK_RING_FIRST) rg_prefetch_wait();
#endif
            if (nround >= BSK_RING_FIRST) round(i0 - W);
            pm.template issue_next<0>(i0 + W);
        }
#if BSK_RING_WAVES >= 3
        if (nround < BSK_RING_FIRST) rg_prefetch_wait();  // a short unit: no round has run, nothing was stored yet
#endif
        if (last_par) pm.template drain<1>(i0 - W);
        else pm.template drain<0>(i0 - W);
        const u32 cnt = ok ? pm.c : 0u;
        const u32 lim = cnt < rows ? cnt : rows;
        lost |= __builtin_amdgcn_ballot_w64(cnt - g >= (u32)LY::R);
        // Reads this kernel cannot finish go to the list of READS for the exact 64-bit machine (k_minimizer_dense<W, true>: 64 per
        // wavefront, per-read slabs in the overflow region): a key tie in one of their min operations (on real data mostly
        // low-complexity reads), more tuples than the slab has rows, or a lane that ran a whole ring ahead of its flushed rows.
        u64 redo = __builtin_amdgcn_ballot_w64(ok && (pm.tmin < 32u || cnt > rows)) | (lost & __builtin_amdgcn_ballot_w64(ok));
#if defined(RG_NOSTAGE) || defined(RG_NOTIE) || defined(RG_NOSTORE) || defined(RG_NOFB)
        redo = 0;
#endif
        {  // what is still in the ring: the lanes' catch-up rows below the frontier, then the groups from the frontier to the unit's last row
            catch_up(lim < T ? lim : T, 1u);
            const u32 cmax = wave_max_u32(lim);
            for (; T < cmax; T += (u32)LY::G) group(T, std::integral_constant<int, 1>{}, (int)(lim - T), 0);
        }
        if (redo) {
            list_append(a, a.rlist, lseg, lcur, redo, lane, r);
        }
#if BSK_RING_WAVES >= 3
        RgWords<NQ> pw_n1;
        u64 d_n2;
        u32 rfl;
        rg_prefetch_take(pw_n1, d_n2, rfl);
#endif
        d_cur = d_n1;
        pw_cur = pw_n1;
        d_n1 = d_n2;
        have = nxt;
        if (r < a.n) {
            if (!((redo >> lane) & 1)) a.refs[ro] = BSK_REF_ROWS | ((base + (u64)lane) << 24) | cnt;  // (listed reads: the list pass writes theirs)
            u8 sbyte = ok ? BSK_ST_OK : BSK_ST_SHORT;
            if (ok && a.rflags) sbyte |= (u8)rfl;
            a.status[ro] = sbyte;
        }
    }
    list_close(a.rlist, lseg, lcur, lane);
}

#ifdef BSK_IMPL_RING
#ifndef BSK_RING_WS
#define BSK_RING_WS(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13)
#endif
bool ring_minimizer_supported(int w) { return w >= 2 && w <= 13; }
u32 ring_minimizer_short_bases() { return 16u * (4 * BSK_RING_NQ - 1); }  // reads up to this length never load inside the k-mer loop
int ring_minimizer_blocks_per_cu(int w) {
    int nb = 0;
    hipError_t e = hipErrorInvalidValue;
    switch (w) {
#define X(WW) \
    case WW: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_minimizer_ring<WW, BSK_RING_NQ, false>, 64, 0); break;
        BSK_RING_WS(X)
#undef X
        default: break;
    }
    if (e != hipSuccess || nb < 1) {
        (void)hipGetLastError();
        nb = 1;
    }
    return nb;
}
void ring_minimizer_launch(int w, bool long_reads, int grid, hipStream_t stream, const KArgs &a) {
    switch (w) {
#define X(WW)                                                                                                          \
    case WW:                                                                                                           \
        if (long_reads) {                                                                                              \
            hipLaunchKernelGGL((k_minimizer_ring<WW, BSK_RING_NQ, true>), dim3(grid), dim3(64), 0, stream, a);         \
        } else {                                                                                                       \
            hipLaunchKernelGGL((k_minimizer_ring<WW, BSK_RING_NQ, false>), dim3(grid), dim3(64), 0, stream, a);        \
        }                                                                                                              \
        hipLaunchKernelGGL((k_minimizer_dense<WW, true>), dim3(grid), dim3(64), 0, stream, a);                         \
        break;
        BSK_RING_WS(X)
#undef X
        default: break;
    }
}
#endif  // BSK_IMPL_RING

}  // namespace bsk
