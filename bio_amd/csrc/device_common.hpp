// device_common.hpp -- gfx950 device helpers shared by the biosketch kernels.
// wave = 64 lanes; every kernel here is launched with 64-thread workgroups
// (one wavefront per workgroup) so "wave" and "block" coincide and
// __syncthreads() is a single-wave barrier + LDS fence.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned short u16;
typedef unsigned char u8;

#define WAVE 64

// ntHash-1 seeds (github.com/will-rowe/nthash v0.4.0 as called from
// sketches/iterator.go:649,659 and sketches/sketch.go:120,179,184; values pinned by
// sketches/sketch_test.go:67-72).  Index = 2-bit code A0 C1 G2 T3 (sketches/kmers.go:23-40).
#define SEED_A 0x3c8bfbb395c60474ULL
#define SEED_C 0x3193c18562a02b4cULL
#define SEED_G 0x20323ed082572324ULL
#define SEED_T 0x295549f54be24456ULL

__host__ __device__ __forceinline__ u64 rol64(u64 v, unsigned n) {
    n &= 63u;
    return n ? (v << n) | (v >> (64u - n)) : v;
}
__host__ __device__ __forceinline__ u64 ror64(u64 v, unsigned n) {
    n &= 63u;
    return n ? (v >> n) | (v << (64u - n)) : v;
}
__device__ __forceinline__ u64 rol1(u64 v) { return (v << 1) | (v >> 63); }
__device__ __forceinline__ u64 ror1(u64 v) { return (v >> 1) | (v << 63); }

__host__ __device__ __forceinline__ u64 seed_fwd_code(unsigned c) {
    return c == 0 ? SEED_A : c == 1 ? SEED_C : c == 2 ? SEED_G : SEED_T;
}
__host__ __device__ __forceinline__ u64 seed_rev_code(unsigned c) {  // seed of the complement base
    return c == 0 ? SEED_T : c == 1 ? SEED_G : c == 2 ? SEED_C : SEED_A;
}
// full byte tables of ntHash-1 (forward: ACGTU + lower case, else 0; reverse: table[b & 7])
__host__ __device__ __forceinline__ u64 seed_fwd_byte(unsigned b) {
    switch (b) {
        case 'A': case 'a': return SEED_A;
        case 'C': case 'c': return SEED_C;
        case 'G': case 'g': return SEED_G;
        case 'T': case 't': case 'U': case 'u': return SEED_T;
        default: return 0;
    }
}
__host__ __device__ __forceinline__ u64 seed_rev_byte(unsigned b) {
    switch (b & 7u) {
        case 1: return SEED_T;
        case 3: return SEED_G;
        case 4: case 5: return SEED_A;
        case 7: return SEED_C;
        default: return 0;
    }
}
__host__ __device__ __forceinline__ unsigned acgt_code(unsigned b) {  // 0..3, or 4 if not ACGTacgt
    switch (b) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 4;
    }
}

// ---- wave primitives ---------------------------------------------------------
// LDS hand-off between lanes of ONE wavefront (every kernel here runs one wavefront per workgroup).
// The LDS unit performs a wave's DS instructions in issue order, so a ds_read issued after another lane's
// ds_write sees it; what is needed is only that the COMPILER keeps the order.  __syncthreads() would also
// emit s_waitcnt vmcnt(0): every hand-off would wait for all outstanding global stores of the wave
// (measured: the tuple copy-out and the stream flushes were store-latency bound because of it).
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// Wave-wide scans / reductions of 32-bit values on DPP moves (row_shr 1, 2, 4, 8 inside the rows of 16 lanes, then row_bcast:15 into
// rows 1 and 3 and row_bcast:31 into rows 2 and 3): six VALU instructions and no LDS.  The ds_bpermute (__shfl_up) form was six
// DEPENDENT trips through the LDS crossbar -- the flush of the per-read-slab kernels scans once per round (every ten steps of the
// protein minimizer), and the scan alone was a third of a round.
#define BSK_DPP0(v, ctrl, rmask) ((u32)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (rmask), 0xf, false))
__device__ __forceinline__ u32 wave_incl_scan_u32(u32 v, int lane) {
    (void)lane;
    v += BSK_DPP0(v, 0x111, 0xf);  // row_shr:1 (lanes without a source add 0)
    v += BSK_DPP0(v, 0x112, 0xf);
    v += BSK_DPP0(v, 0x114, 0xf);
    v += BSK_DPP0(v, 0x118, 0xf);
    v += BSK_DPP0(v, 0x142, 0xa);  // row_bcast:15 -> rows 1, 3
    v += BSK_DPP0(v, 0x143, 0xc);  // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ u64 wave_incl_scan_u64(u64 v, int lane) {  // the same ladder on two words with the carry between them
    (void)lane;
    u32 lo = (u32)v, hi = (u32)(v >> 32);
#define BSK_SCAN64_STEP(ctrl, rmask)                      \
    {                                                     \
        const u32 dl = BSK_DPP0(lo, ctrl, rmask), dh = BSK_DPP0(hi, ctrl, rmask); \
        const u32 nl = lo + dl;                           \
        hi += dh + (nl < lo ? 1u : 0u);                   \
        lo = nl;                                          \
    }
    BSK_SCAN64_STEP(0x111, 0xf)
    BSK_SCAN64_STEP(0x112, 0xf)
    BSK_SCAN64_STEP(0x114, 0xf)
    BSK_SCAN64_STEP(0x118, 0xf)
    BSK_SCAN64_STEP(0x142, 0xa)
    BSK_SCAN64_STEP(0x143, 0xc)
#undef BSK_SCAN64_STEP
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 wave_sum_u64(u64 v) {  // lane 63 of the scan, in every lane
    const u64 s = wave_incl_scan_u64(v, 0);
    return ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(s >> 32), 63) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)s, 63);
}
__device__ __forceinline__ u32 wave_max_u32(u32 v) {  // the same ladder with max (0 is the identity of an unsigned maximum), read from lane 63
    u32 t;
    t = BSK_DPP0(v, 0x111, 0xf); v = t > v ? t : v;
    t = BSK_DPP0(v, 0x112, 0xf); v = t > v ? t : v;
    t = BSK_DPP0(v, 0x114, 0xf); v = t > v ? t : v;
    t = BSK_DPP0(v, 0x118, 0xf); v = t > v ? t : v;
    t = BSK_DPP0(v, 0x142, 0xa); v = t > v ? t : v;
    t = BSK_DPP0(v, 0x143, 0xc); v = t > v ? t : v;
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ u32 wave_bcast_u32(u32 v, int src) { return (u32)__builtin_amdgcn_readlane((int)v, src); }  // src is wave-uniform
__device__ __forceinline__ u64 wave_bcast_u64(u64 v, int src) {
    return ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), src) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)v, src);
}

// ---- work distribution: persistent waves pull units (64 reads) from a ticket ----
// Tickets are handed out in program order of the atomic, so every unit with a
// smaller ticket is already running or done: the look-back below cannot deadlock,
// whatever the workgroup -> CU/XCD placement.
__device__ __forceinline__ u32 next_ticket(u32 *ticket, int lane) {
    u32 t = 0;
    if (lane == 0) t = atomicAdd(ticket, 1u);
    return (u32)__builtin_amdgcn_readfirstlane((int)t);
}

// ---- tickets from eight heads ---------------------------------------------------------
// ONE word hands out ~88 tickets a microsecond however many waves pull (MI355X guide, "dequeue": a returning device-scope atomicAdd on one
// head word saturates there).  Kernels whose units are many thousand bases long and come several to a ticket never notice; a kernel that must
// take its units ONE by one, in order (a look-back over the units: several units to a ticket would chain the tickets), is bound by exactly
// that: 240 000 units of 64 tiles = 2.7 ms of k_tile_stitch's 2.8.  Eight heads, one per XCD and a cache line each: head x hands out the
// units x, x + 8, x + 16, ...; a wave pulls from its own XCD's head and, when that has run out, goes round the others (nothing says every XCD
// holds a wave of the launch).  The units still start in (nearly) rising order, which is all a decoupled look-back asks for.
__device__ __forceinline__ u32 xcc_id() {
    u32 x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
    return x & 7u;
}
constexpr u32 HEAD_WORDS = 8 * 32;  // u32 words of the eight heads (zeroed before the launch)
// a look-back scratch of nunits granules with the heads behind it (u64 words): where the heads start, and the whole of it
__host__ __device__ constexpr size_t lb_heads_at(u32 nunits) { return ((size_t)nunits + 15u) & ~(size_t)15u; }
__host__ __device__ constexpr size_t lb_words_with_heads(u32 nunits) { return lb_heads_at(nunits) + HEAD_WORDS / 2; }
struct HeadTickets {
    u32 *heads;
    u32 x0, hh;
    __device__ __forceinline__ HeadTickets(u32 *h) : heads(h), x0(xcc_id()), hh(0) {}
    __device__ __forceinline__ u32 head() const { return (x0 + hh) & 7u; }
    // the next unit (< nunits), or ~0u when every head has run out
    __device__ __forceinline__ u32 next(u32 nunits, int lane) {
        while (hh < 8u) {
            const u32 hx = head();
            const u32 u = next_ticket(heads + hx * 32u, lane) * 8u + hx;
            if (u < nunits) return u;
            ++hh;
        }
        return ~0u;
    }
};

// ---- decoupled look-back: exclusive prefix of per-unit tuple totals ---------------
// One 8-byte granule per unit: bits 63..62 state (0 empty, 1 aggregate, 2 inclusive),
// bits 61..0 value.  Granules are written/read with relaxed agent-scope atomics
// (sc1 stores/loads: they bypass the non-coherent per-XCD L2s); the granule IS the
// payload, so no fence is needed (MI355X guide, G16 R2).
#define LB_AGG (1ULL << 62)
#define LB_INC (2ULL << 62)
#define LB_VAL ((1ULL << 62) - 1)

__device__ __forceinline__ void lb_store(u64 *p, u64 v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 lb_load(u64 *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ u64 lookback_exclusive(u64 *st, u32 unit, u64 total, int lane) {
    if (unit == 0) {
        if (lane == 0) lb_store(&st[0], LB_INC | total);
        return 0;
    }
    if (lane == 0) lb_store(&st[unit], LB_AGG | total);
    u64 excl = 0;
    long long end = (long long)unit - 1;
    for (;;) {
        long long idx = end - lane;
        u64 s = idx >= 0 ? lb_load(&st[idx]) : LB_INC;  // virtual unit -1: inclusive 0
        u32 state = (u32)(s >> 62);
        u64 inc_mask = __ballot(state == 2);
        u64 empty_mask = __ballot(state == 0);
        if (inc_mask) {
            int first = __builtin_ctzll(inc_mask);
            u64 needed = first == 63 ? ~0ULL : ((2ULL << first) - 1);
            if (empty_mask & needed) {
                __builtin_amdgcn_s_sleep(2);
                continue;
            }
            excl += wave_sum_u64(lane <= first ? (s & LB_VAL) : 0ULL);
            break;
        }
        if (empty_mask) {
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        excl += wave_sum_u64(s & LB_VAL);
        end -= 64;
    }
    if (lane == 0) lb_store(&st[unit], LB_INC | (excl + total));
    return excl;
}

// The same walk by a unit that only wants to know where `unit` begins (every granule before it is somebody else's to publish).
__device__ __forceinline__ u64 lookback_peek(u64 *st, u32 unit, int lane) {
    u64 excl = 0;
    long long end = (long long)unit - 1;
    for (;;) {
        long long idx = end - lane;
        u64 s = idx >= 0 ? lb_load(&st[idx]) : LB_INC;
        u32 state = (u32)(s >> 62);
        u64 inc_mask = __ballot(state == 2);
        u64 empty_mask = __ballot(state == 0);
        if (inc_mask) {
            int first = __builtin_ctzll(inc_mask);
            u64 needed = first == 63 ? ~0ULL : ((2ULL << first) - 1);
            if (empty_mask & needed) {
                __builtin_amdgcn_s_sleep(2);
                continue;
            }
            excl += wave_sum_u64(lane <= first ? (s & LB_VAL) : 0ULL);
            break;
        }
        if (empty_mask) {
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        excl += wave_sum_u64(s & LB_VAL);
        end -= 64;
    }
    return excl;
}

// splitmix64: counter-based generator for bsk_batch_synth
__host__ __device__ __forceinline__ u64 splitmix64(u64 x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
