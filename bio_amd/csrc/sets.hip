// sets.hip -- sketch SETS: the distinct hash values of a result, ascending, per sequence or for the whole batch,
// optionally FracMinHash-filtered (hash <= MaxUint64/scale, the rule of iterator.go:181-185).  This is the form the
// downstream consumers of the reference's sketches keep on disk (kmcp / unikmer: sorted unique uint64 lists):
// collecting Next() values into a slice, sort, de-duplicate.  Done on the device with rocPRIM (segmented radix sort,
// scan) so that only the final sets cross PCIe.
#include <chrono>
#include <cstdio>
#include <hip/hip_runtime.h>
#include <string.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include <algorithm>
#include <new>
#include <vector>

#include "biosketch.h"
#include "host_types.hpp"

struct bsk_sets {
    bsk_ctx *ctx = nullptr;
    u64 n_sets = 0, n_values = 0;
    u64 *offsets = nullptr;  // [n_sets + 1]
    u64 *values = nullptr;   // [n_values] ascending inside a set
    size_t c_offsets = 0, c_values = 0;  // bytes allocated (grow-only when the object is re-used: bsk_result_sets_reuse)
};

namespace {

// where the values of sequence r are: packed reference word or the wide arrays (host_types.hpp)
// stride: 1, or 64 for a read of a unit written as rows (BSK_REF_ROWS, kernels_ring.hpp)
__device__ __forceinline__ void seq_span(const u64 *refs, const u64 *wfirst, const u64 *wcount, u64 r, u64 &first, u64 &cnt, u64 &stride) {
    if (refs) {
        first = BSK_REF_FIRST(refs[r]);
        cnt = BSK_REF_COUNT(refs[r]);
        stride = BSK_REF_STRIDE(refs[r]);
    } else {
        first = wfirst[r];
        cnt = wcount[r];
        stride = 1;
    }
}

// dense copy of every sequence's values (one wavefront per sequence); values above maxhash become the sentinel ~0 so that
// they sort to the end of their segment
__global__ void k_gather_values(const u64 *hash, const u64 *refs, const u64 *wfirst, const u64 *wcount, const u64 *dst, u64 n, u64 maxhash,
                                u64 *out) {
    // a group of 16 lanes per sequence (short reads hold two dozen values: a whole wavefront per sequence left most lanes idle)
    const u64 grp = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 4, ng = ((u64)gridDim.x * blockDim.x) >> 4;
    for (u64 r = grp; r < n; r += ng) {
        u64 f, c, st;
        seq_span(refs, wfirst, wcount, r, f, c, st);
        const u64 d = dst[r];
        for (u64 t = threadIdx.x & 15; t < c; t += 16) {
            const u64 h = hash[f + t * st];
            out[d + t] = h > maxhash ? ~0ULL : h;
        }
    }
}

__global__ void k_mark_heads(const u64 *offs, u64 n_sets, u8 *head) {
    for (u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x; s < n_sets; s += (u64)gridDim.x * blockDim.x)
        if (offs[s + 1] > offs[s]) head[offs[s]] = 1;
}

// keep[i] = 1 iff value i is the first occurrence inside its set and passes the filter
__global__ void k_flag_unique(const u64 *v, const u8 *head, u64 n, u64 maxhash, u32 *keep) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
        keep[i] = (v[i] <= maxhash && (head[i] || v[i] != v[i - 1])) ? 1u : 0u;
}

__global__ void k_scatter_unique(const u64 *v, const u32 *keep, const u64 *pos, u64 n, u64 *out) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
        if (keep[i]) out[pos[i]] = v[i];
}

// new offsets: position of the first value of every set in the compacted array
__global__ void k_new_offsets(const u64 *offs_in, const u64 *pos, u64 n_sets, u64 n_in, u64 n_out, u64 *offs_out) {
    for (u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x; s <= n_sets; s += (u64)gridDim.x * blockDim.x)
        offs_out[s] = offs_in[s] < n_in ? pos[offs_in[s]] : n_out;
}

// ---- small sets: every sequence holds at most 64 values (short reads: ~22 minimizers, ~7 syncmers per 150-bp read) ----
// A segmented radix sort spends eight digit passes on segments of two dozen keys.  Here a sequence is sorted by ONE ROW of 16
// lanes (a DPP row: four sequences per wavefront), every lane holding E = 2 or 4 of its values (element e = reg * 16 + lane):
// a bitonic network whose exchange distances 1, 2, 4, 8 are DPP moves inside the row (quad_perm, row_half_mirror, row_ror:8 --
// no LDS crossbar: the first version, one value per lane and ds_bpermute for every distance, was bound by it at 5.4 ms per 10^7
// reads) and whose distances 16, 32 are compare-exchanges between a lane's own registers.  A wavefront takes the 32-value
// network unless one of its four sequences holds more than 32 values (0.3 % of 150-bp reads at k = 21, w = 11), the 16-value
// network when none holds more than 16 (syncmers).  Duplicates go by comparing with the element below; the distinct values
// leave as one dense run at the sequence's input offset (ballot + bit counts), k_move_seqs then shifts the runs to their final
// offsets.  (Measured and dropped: ONE kernel that stages a unit of 32 sequences in LDS, takes its final offset from a
// decoupled look-back and writes the final array directly -- no dense copy, no second scan, no move, half the traffic -- ran
// 4.4 ms against the 2.8 ms of rows + move: at 9 waves per CU, the 16 KB of staging, a unit's ticket, reference words, values
// and look-back are five exposed round trips per 32 sequences.)
#define SMALL_CAP 64
template <int CTRL>
__device__ __forceinline__ u32 dpp_mov(u32 x) {
    return (u32)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, 0xf, 0xf, false);
}
template <int J>
__device__ __forceinline__ u64 row_xor(u64 v) {  // lane l of every row of 16 receives lane l ^ J
    u32 lo = (u32)v, hi = (u32)(v >> 32);
    if (J == 1) {
        lo = dpp_mov<0xB1>(lo);  // quad_perm [1,0,3,2]
        hi = dpp_mov<0xB1>(hi);
    } else if (J == 2) {
        lo = dpp_mov<0x4E>(lo);  // quad_perm [2,3,0,1]
        hi = dpp_mov<0x4E>(hi);
    } else if (J == 4) {
        lo = dpp_mov<0x1B>(dpp_mov<0x141>(lo));  // row_half_mirror (l -> 7 - l within 8), then quad_perm [3,2,1,0]
        hi = dpp_mov<0x1B>(dpp_mov<0x141>(hi));
    } else {
        lo = dpp_mov<0x128>(lo);  // row_ror:8
        hi = dpp_mov<0x128>(hi);
    }
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 row_below(u64 v) {  // lane l receives lane (l - 1) & 15 of its row
    return ((u64)dpp_mov<0x121>((u32)(v >> 32)) << 32) | dpp_mov<0x121>((u32)v);  // row_ror:1
}

// ascending bitonic sort of the 16 * E values of every row (element e = reg * 16 + l)
template <int E>
__device__ __forceinline__ void sort_rows(u64 (&v)[E], int l) {
    constexpr int NE = 16 * E;
#pragma unroll
    for (int k = 2; k <= NE; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 16) {  // partner in another register of the same lane
                const int dj = j >> 4;
#pragma unroll
                for (int reg = 0; reg < E; ++reg) {
                    if (reg & dj) continue;
                    const bool asc = ((reg * 16) & k) == 0;
                    const u64 a = v[reg], b = v[reg | dj];
                    const bool lt = b < a;
                    const u64 mn = lt ? b : a, mx = lt ? a : b;
                    v[reg] = asc ? mn : mx;
                    v[reg | dj] = asc ? mx : mn;
                }
            } else {
                const bool lower = (l & j) == 0;
#pragma unroll
                for (int reg = 0; reg < E; ++reg) {
                    const bool asc = k < 16 ? (l & k) == 0 : ((reg * 16) & k) == 0;
                    const bool take_min = lower == asc;
                    u64 o;
                    switch (j) {
                        case 1: o = row_xor<1>(v[reg]); break;
                        case 2: o = row_xor<2>(v[reg]); break;
                        case 4: o = row_xor<4>(v[reg]); break;
                        default: o = row_xor<8>(v[reg]); break;
                    }
                    const bool o_less = o < v[reg];
                    v[reg] = (take_min == o_less) ? o : v[reg];
                }
            }
        }
    }
}

// load, filter, sort and flag the distinct values of the row's sequence: keep[reg] / rank rk[reg] of element reg * 16 + l among
// the kept ones; returns their number (the same in all 16 lanes of the row)
template <int E>
__device__ __forceinline__ u32 row_set(const u64 *src, u32 stride, u32 c, u64 maxhash, int l, int row, u64 (&v)[E], bool (&keep)[E], u32 (&rk)[E]) {
    u32 nvalid = 0;
#pragma unroll
    for (int reg = 0; reg < E; ++reg) {
        const u32 e = (u32)(reg * 16 + l);
        u64 x = ~0ULL;
        bool valid = false;
        if (e < c) {
            x = src[(size_t)e * stride];
            valid = x <= maxhash;
            if (!valid) x = ~0ULL;  // filtered values sort to the end with the padding
        }
        v[reg] = x;
        const u64 vm = __ballot(valid);
        nvalid += (u32)__builtin_popcount((u32)(vm >> (row * 16)) & 0xffffu);
    }
    sort_rows<E>(v, l);
    u32 base = 0;
#pragma unroll
    for (int reg = 0; reg < E; ++reg) {
        const u64 t = (reg > 0 && l == 15) ? v[reg - 1] : v[reg];  // lane 0 looks at lane 15 of the register below
        const u64 below = row_below(t);
        const u32 e = (u32)(reg * 16 + l);
        keep[reg] = e < nvalid && (e == 0 || v[reg] != below);
        const u32 bits = (u32)(__ballot(keep[reg]) >> (row * 16)) & 0xffffu;
        rk[reg] = base + (u32)__builtin_popcount(bits & ((1u << l) - 1u));
        base += (u32)__builtin_popcount(bits);
    }
    return base;
}

template <int E>
__device__ __forceinline__ u32 rows_store(const u64 *src, u32 stride, u32 c, u64 maxhash, int l, int row, u64 *dst) {
    u64 v[E];
    bool keep[E];
    u32 rk[E];
    const u32 u = row_set<E>(src, stride, c, maxhash, l, row, v, keep, rk);
#pragma unroll
    for (int reg = 0; reg < E; ++reg)
        if (keep[reg]) dst[rk[reg]] = v[reg];
    return u;
}
__global__ __launch_bounds__(256) void k_sets_rows(const u64 *hash, const u64 *refs, const u64 *wfirst, const u64 *wcount, const u64 *offs, u64 n,
                                                   u64 maxhash, u64 *tmp, u64 *ucount) {
    const int lane = threadIdx.x & 63, l = lane & 15, row = lane >> 4;
    const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((u64)gridDim.x * blockDim.x) >> 6;
    const u64 nit = (n + 3) / 4;  // wave iterations: four sequences each
    for (u64 it = wave; it < nit; it += nw) {
        const u64 r = it * 4 + (u64)row;
        u64 first = 0, c = 0, d = 0, st = 1;
        if (r < n) {
            seq_span(refs, wfirst, wcount, r, first, c, st);
            d = offs[r];
        }
        u32 u;
        if (__ballot(c > 32)) u = rows_store<4>(hash + first, (u32)st, (u32)c, maxhash, l, row, tmp + d);
        else if (__ballot(c > 16)) u = rows_store<2>(hash + first, (u32)st, (u32)c, maxhash, l, row, tmp + d);
        else u = rows_store<1>(hash + first, (u32)st, (u32)c, maxhash, l, row, tmp + d);
        if (l == 0 && r < n) ucount[r] = u;
    }
}

// ---- exclusive scan of n per-sequence numbers (counts taken from the result's reference words, or an array) into out[0 .. n]
// (out[n] = the total); three small kernels: block sums, one block over the block sums, local scan + block offset ----
struct CountOf {
    const u64 *refs, *wfirst, *wcount;
    __device__ __forceinline__ u64 operator()(u64 r) const {
        u64 f, c, st;
        seq_span(refs, wfirst, wcount, r, f, c, st);
        return c;
    }
};
struct ArrayOf {
    const u64 *a;
    __device__ __forceinline__ u64 operator()(u64 r) const { return a[r]; }
};
struct KeepOf {
    const u32 *a;
    __device__ __forceinline__ u64 operator()(u64 r) const { return (u64)a[r]; }
};
#define SCAN_PER_THREAD 8
#define SCAN_BLOCK 256
#define SCAN_CHUNK (SCAN_PER_THREAD * SCAN_BLOCK)
__device__ __forceinline__ u64 wave_incl(u64 x, int lane) {
    for (int d = 1; d < 64; d <<= 1) {
        const u64 t = __shfl_up(x, d, 64);
        if (lane >= d) x += t;
    }
    return x;
}
template <class F>
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_sums(F f, u64 n, u64 *part, u64 *mx) {
    __shared__ u64 s_w[SCAN_BLOCK / 64];
    const u64 b0 = (u64)blockIdx.x * SCAN_CHUNK;
    u64 sum = 0, m = 0, x[SCAN_PER_THREAD];
#pragma unroll
    for (int i = 0; i < SCAN_PER_THREAD; ++i) {  // all loads in flight before the first use
        const u64 r = b0 + (u64)i * SCAN_BLOCK + threadIdx.x;
        x[i] = r < n ? f(r) : 0;
    }
#pragma unroll
    for (int i = 0; i < SCAN_PER_THREAD; ++i) {
        sum += x[i];
        m = x[i] > m ? x[i] : m;
    }
    for (int d = 32; d; d >>= 1) {
        sum += __shfl_xor(sum, d, 64);
        const u64 t = __shfl_xor(m, d, 64);
        m = t > m ? t : m;
    }
    __shared__ u64 s_m[SCAN_BLOCK / 64];
    if ((threadIdx.x & 63) == 0) {
        s_w[threadIdx.x >> 6] = sum;
        s_m[threadIdx.x >> 6] = m;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 t = 0, bm = 0;
        for (int w = 0; w < SCAN_BLOCK / 64; ++w) {
            t += s_w[w];
            bm = s_m[w] > bm ? s_m[w] : bm;
        }
        part[blockIdx.x] = t;
        // one atomic per block, and only when it can raise the maximum (thousands of blocks hitting one address serialise in the L2)
        if (mx && bm > __hip_atomic_load(mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax((unsigned long long *)mx, (unsigned long long)bm);
    }
}
__global__ __launch_bounds__(1024) void k_scan_top(u64 *part, u64 nb, u64 *total) {  // in place: part[b] <- sum of part[0 .. b)
    __shared__ u64 s_w[16];
    __shared__ u64 s_carry;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (u64 b0 = 0; b0 < nb; b0 += 1024) {
        const u64 i = b0 + threadIdx.x;
        const u64 x = i < nb ? part[i] : 0;
        const u64 inc = wave_incl(x, lane);
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        u64 before = s_carry;
        for (int q = 0; q < w; ++q) before += s_w[q];
        if (i < nb) part[i] = before + inc - x;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = s_carry;
}
template <class F>
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_apply(F f, u64 n, const u64 *part, const u64 *total, u64 *out) {
    __shared__ u64 s_w[SCAN_BLOCK / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // thread t owns SCAN_PER_THREAD consecutive numbers
    const u64 r0 = (u64)blockIdx.x * SCAN_CHUNK + (u64)threadIdx.x * SCAN_PER_THREAD;
    u64 x[SCAN_PER_THREAD], sum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_PER_THREAD; ++i) x[i] = r0 + i < n ? f(r0 + i) : 0;
#pragma unroll
    for (int i = 0; i < SCAN_PER_THREAD; ++i) sum += x[i];
    const u64 inc = wave_incl(sum, lane);
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    u64 run = part[blockIdx.x] + inc - sum;
    for (int q = 0; q < w; ++q) run += s_w[q];
#pragma unroll
    for (int i = 0; i < SCAN_PER_THREAD; ++i) {
        if (r0 + i < n) out[r0 + i] = run;
        run += x[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = *total;
}
// out[0..n] = exclusive scan of f(0..n-1); *total_dev (device) receives the total, *mx_dev (may be NULL) the maximum
template <class F>
hipError_t scan_counts(hipStream_t st, F f, u64 n, u64 *part, u64 *out, u64 *total_dev, u64 *mx_dev) {
    const u64 nb = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    if (nb) hipLaunchKernelGGL(k_scan_sums<F>, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, f, n, part, mx_dev);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, st, part, nb, total_dev);
    if (nb) hipLaunchKernelGGL(k_scan_apply<F>, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, f, n, part, total_dev, out);
    else hipLaunchKernelGGL(k_scan_apply<F>, dim3(1), dim3(SCAN_BLOCK), 0, st, f, n, part, total_dev, out);
    return hipGetLastError();
}

// final placement: sequence r's run [in_off[r], +ucount[r]) -> [out_off[r], ...); a group of 32 lanes per sequence
__global__ void k_move_seqs(const u64 *tmp, const u64 *in_off, const u64 *out_off, const u64 *ucount, u64 n, u64 *out) {
    const u64 g = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 5, ng = ((u64)gridDim.x * blockDim.x) >> 5;
    const u32 gl = threadIdx.x & 31;
    for (u64 r = g; r < n; r += ng) {
        const u64 s0 = in_off[r], d0 = out_off[r], t = ucount[r];
        for (u64 i = gl; i < t; i += 32) out[d0 + i] = tmp[s0 + i];
    }
}
// the same placement by GROUPS of 64 sequences (see k_gather_groups below): the group's output range is contiguous, the lanes take
// its elements in order and find each element's sequence by a six-step search over the lanes' output offsets (ds_bpermute)
__global__ __launch_bounds__(256) void k_move_groups(const u64 *tmp, const u64 *in_off, const u64 *out_off, const u64 *ucount, u64 n, u64 *out) {
    const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((u64)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    const u64 ngroups = (n + 63) / 64;
    for (u64 g = wave; g < ngroups; g += nw) {
        const u64 r = g * 64 + lane;
        const u64 d = out_off[r < n ? r : n];
        const u64 src = r < n ? in_off[r] : 0;
        const u32 cnt = r < n ? (u32)ucount[r] : 0u;
        const u64 d0 = wave_bcast_u64(d, 0);
        const u32 dl = (u32)(d - d0);
        const u32 T = (u32)__builtin_amdgcn_readlane((int)(dl + cnt), 63);
        const u32 slo = (u32)src, shi = (u32)(src >> 32);
        for (u32 j0 = 0; j0 < T; j0 += 128) {
            u32 j[2], own[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                j[q] = j0 + 64 * q + (u32)lane;
                own[q] = 0;
            }
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const u32 probe = own[q] + (u32)s;
                    const u32 v = (u32)__builtin_amdgcn_ds_bpermute((int)(probe << 2), (int)dl);
                    own[q] = v <= j[q] ? probe : own[q];
                }
            }
            u64 hv[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int a = (int)(own[q] << 2);
                const u32 t = j[q] - (u32)__builtin_amdgcn_ds_bpermute(a, (int)dl);
                const u32 b0 = (u32)__builtin_amdgcn_ds_bpermute(a, (int)slo), b1 = (u32)__builtin_amdgcn_ds_bpermute(a, (int)shi);
                hv[q] = j[q] < T ? tmp[(((u64)b1 << 32) | b0) + t] : 0;
            }
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (j[q] < T) out[d0 + j[q]] = hv[q];
        }
    }
}
int grid_of(bsk_ctx *ctx, u64 items, int block) {
    const u64 g = (items + block - 1) / block;
    return (int)std::max<u64>(1, std::min<u64>(g, (u64)ctx->cus * 16));
}

}  // namespace

// Dense device copy of a result's tuples: the sketch kernels leave gaps between the 64-sequence units' slabs, a device-side consumer
// wants CSR.  Offsets by the scan above, tuples by one wavefront per sequence; everything in the context's grow-only pool.
namespace {
__global__ void k_compact(const u64 *hash, const u32 *pos, const u64 *refs, const u64 *wfirst, const u64 *wcount, const u64 *dstoff, u64 n,
                          u64 *ohash, u32 *opos) {
    const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 r = wave; r < n; r += nw) {
        u64 b, cnt, st;
        seq_span(refs, wfirst, wcount, r, b, cnt, st);
        const u64 d = dstoff[r];
        for (u64 t = threadIdx.x & 63; t < cnt; t += 64) {
            ohash[d + t] = hash[b + t * st];
            if (opos) opos[d + t] = pos[b + t * st];
        }
    }
}

// The same copy for results with packed reference words (every batch of reads): one wavefront per GROUP of 64 consecutive sequences.
// The group's output range [dstoff[g*64], dstoff[g*64+64)) is contiguous, so the lanes take its elements in order -- whole lines of
// coalesced stores, whatever the sequences' counts -- and find each element's sequence by a six-step search over the lanes' own
// offsets (ds_bpermute: no LDS allocation, no table to build); two rows of 64 elements per trip, so that four loads are in flight.
// (One wavefront per sequence kept 22 of 64 lanes busy on 150-base reads and ran at 1.9 TB/s: a dense copy cost 2.4 sketches.)
// NARROW: bsk_result_fetch_narrow's outputs (u16 positions, u32 offsets, the 15-bit check).
// ROWS (results of k_minimizer_ring, launched with one wavefront per workgroup): a group whose sequences are stored as unit rows
// (element t of a sequence 64 tuples after element t - 1) is read ROW BY ROW -- the lanes are the sequences, one coalesced load per
// row -- into a 32 KB LDS image of the group's output range, which then leaves in order; hashes first, positions through the same
// image.  (Elements in output order are 512 bytes apart in such a slab: 64 lines per load instruction, 1.6 TB/s.)  Groups of more than
// 4 096 tuples take the search below.
#define BSK_GATHER_LCAP 4096
template <bool NARROW, bool ROWS = false>
__global__ __launch_bounds__(256) void k_gather_groups(const u64 *hash, const u32 *pos, const u64 *refs, const u64 *dstoff, u64 n, u64 *ohash,
                                                       void *opos_, u32 *ooff, u32 *flag) {
    const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((u64)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    const u64 ngroups = (n + 64) / 64;  // (NARROW: entry n of the offsets is written by the group that holds slot n)
    u32 wide = 0;
    for (u64 g = wave; g < ngroups; g += nw) {
        const u64 r = g * 64 + lane;
        const u64 d = dstoff[r < n ? r : n];
        u64 src = 0;
        u32 cnt = 0;
        if (r < n) {
            const u64 ref = refs[r];
            cnt = (u32)BSK_REF_COUNT(ref);
            src = BSK_REF_FIRST(ref) | (BSK_REF_STRIDE(ref) > 1 ? 1ULL << 63 : 0);
        }
        if (NARROW && r <= n) ooff[r] = (u32)d;
        const u64 d0 = wave_bcast_u64(d, 0);
        const u32 dl = (u32)(d - d0);  // < 2^30: 64 sequences of < 2^24 tuples
        const u32 T = (u32)__builtin_amdgcn_readlane((int)(dl + cnt), 63);
        const u32 slo = (u32)src, shi = (u32)(src >> 32);
        if (ROWS && T <= BSK_GATHER_LCAP) {
            // (a listed read of such a unit lies elsewhere with stride 1 -- at 250 bases every second group holds one: the stride is
            // the lane's own, only the group's size sends it to the search below)
            constexpr int GU = 32;  // rows requested together
            __shared__ u64 image[ROWS ? BSK_GATHER_LCAP : 1];
            const u64 base = src & ~(1ULL << 63), step = (src >> 63) ? 64 : 1;
            const u32 maxc = wave_max_u32(cnt);
            for (u32 t0 = 0; t0 < maxc; t0 += GU) {
                u64 v[GU];
#pragma unroll
                for (int i = 0; i < GU; ++i) v[i] = t0 + i < cnt ? hash[base + (u64)(t0 + i) * step] : 0;
#pragma unroll
                for (int i = 0; i < GU; ++i)
                    if (t0 + i < cnt) image[dl + t0 + i] = v[i];
            }
            wave_sync_lds();
            for (u32 j = (u32)lane; j < T; j += 64) ohash[d0 + j] = image[j];
            if (pos) {
                wave_sync_lds();
                u32 *image32 = reinterpret_cast<u32 *>(image);
                for (u32 t0 = 0; t0 < maxc; t0 += GU) {
                    u32 v[GU];
#pragma unroll
                    for (int i = 0; i < GU; ++i) v[i] = t0 + i < cnt ? pos[base + (u64)(t0 + i) * step] : 0;
#pragma unroll
                    for (int i = 0; i < GU; ++i)
                        if (t0 + i < cnt) image32[dl + t0 + i] = v[i];
                }
                wave_sync_lds();
                for (u32 j = (u32)lane; j < T; j += 64) {
                    const u32 q = image32[j];
                    if (NARROW) {
                        wide |= q & 0x7fff8000u;
                        reinterpret_cast<u16 *>(opos_)[d0 + j] = (u16)((q & 0x7fffu) | ((q >> 16) & 0x8000u));
                    } else {
                        reinterpret_cast<u32 *>(opos_)[d0 + j] = q;
                    }
                }
            }
            wave_sync_lds();
            continue;
        }
        for (u32 j0 = 0; j0 < T; j0 += 128) {
            u32 j[2], own[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                j[q] = j0 + 64 * q + (u32)lane;
                own[q] = 0;  // the largest lane whose offset is <= j (sequences without tuples share their successor's offset)
            }
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const u32 probe = own[q] + (u32)s;
                    const u32 v = (u32)__builtin_amdgcn_ds_bpermute((int)(probe << 2), (int)dl);
                    own[q] = v <= j[q] ? probe : own[q];
                }
            }
            u64 hv[2];
            u32 pv[2];
            bool ok[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int a = (int)(own[q] << 2);
                const u32 t = j[q] - (u32)__builtin_amdgcn_ds_bpermute(a, (int)dl);
                const u32 b0 = (u32)__builtin_amdgcn_ds_bpermute(a, (int)slo), b1 = (u32)__builtin_amdgcn_ds_bpermute(a, (int)shi);
                const u64 at = (((u64)(b1 & 0x7fffffffu) << 32) | b0) + ((b1 >> 31) ? (u64)t * 64 : (u64)t);
                ok[q] = j[q] < T;
                hv[q] = ok[q] ? hash[at] : 0;
                pv[q] = (ok[q] && pos) ? pos[at] : 0;
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (!ok[q]) continue;
                ohash[d0 + j[q]] = hv[q];
                if (pos) {
                    if (NARROW) {
                        wide |= pv[q] & 0x7fff8000u;
                        reinterpret_cast<u16 *>(opos_)[d0 + j[q]] = (u16)((pv[q] & 0x7fffu) | ((pv[q] >> 16) & 0x8000u));
                    } else {
                        reinterpret_cast<u32 *>(opos_)[d0 + j[q]] = pv[q];
                    }
                }
            }
        }
    }
    if (NARROW && wide) atomicOr(flag, 1u);
}
}  // namespace

extern "C" int bsk_result_compact(bsk_ctx *ctx, const bsk_result *r, const uint64_t **offsets, const uint64_t **hash, const uint32_t **pos,
                                  uint64_t *n_tuples) {
    if (!ctx || !r || !offsets || !hash) return fail_arg(ctx, "bsk_result_compact: null argument");
    if (r->ctx != ctx) return fail_arg(ctx, "bsk_result_compact: result belongs to another context");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const u64 n = r->n;
    auto pool = [&](int slot, size_t bytes, void **outp) -> hipError_t {
        if (ctx->tmp_cap[slot] < bytes) {
            (void)hipFree(ctx->tmp[slot]);
            ctx->tmp[slot] = nullptr;
            ctx->tmp_cap[slot] = 0;
            const size_t want = bytes + bytes / 4 + 256;
            const hipError_t e = hipMalloc(&ctx->tmp[slot], want);
            if (e != hipSuccess) return e;
            ctx->tmp_cap[slot] = want;
        }
        *outp = ctx->tmp[slot];
        return hipSuccess;
    };
    u64 *offs = nullptr, *part = nullptr, *oh = nullptr;
    u32 *op = nullptr;
    const bool timing = ctx->opt.timing;  // dev: wall time of the phases, to stderr
    const auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (timing) fprintf(stderr, "bsk_result_compact: %-18s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    };
    HIPCHK(ctx, pool(16, (n + 2) * 8, (void **)&offs));
    HIPCHK(ctx, pool(17, ((n + SCAN_CHUNK - 1) / SCAN_CHUNK + 2) * 8, (void **)&part));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_total, 0, 16, st));
    HIPCHK(ctx, scan_counts(st, CountOf{r->refs, r->wfirst, r->wcount}, n, part, offs, ctx->d_total + 1, (u64 *)nullptr));
    u64 T = 0;
    HIPCHK(ctx, hipMemcpyAsync(&T, ctx->d_total + 1, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    lap("offsets scanned");
    HIPCHK(ctx, pool(18, (T + 1) * 8, (void **)&oh));
    if (r->pos) HIPCHK(ctx, pool(19, (T + 1) * 4, (void **)&op));
    lap("arrays");
    if (T) {
        if (r->refs && !ctx->opt.no_group_gather && r->unit_rows)  // unit rows
            hipLaunchKernelGGL((k_gather_groups<false, true>), dim3(grid_of(ctx, n + 64, 1)), dim3(64), 0, st, r->hash, r->pos, r->refs, offs, n, oh, (void *)op,
                               (u32 *)nullptr, (u32 *)nullptr);
        else if (r->refs && !ctx->opt.no_group_gather)
            hipLaunchKernelGGL((k_gather_groups<false, false>), dim3(grid_of(ctx, n + 64, 256)), dim3(256), 0, st, r->hash, r->pos, r->refs, offs, n, oh, (void *)op,
                               (u32 *)nullptr, (u32 *)nullptr);
        else
            hipLaunchKernelGGL(k_compact, dim3(grid_of(ctx, n * 64, 256)), dim3(256), 0, st, r->hash, r->pos, r->refs, r->wfirst, r->wcount, offs, n, oh,
                               r->pos ? op : nullptr);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipStreamSynchronize(st));
    }
    lap("tuples gathered");
    *offsets = (const uint64_t *)offs;
    *hash = (const uint64_t *)oh;
    if (pos) *pos = r->pos ? op : nullptr;
    if (n_tuples) *n_tuples = T;
    return BSK_OK;
}

// bsk_result_fetch with half the side traffic: the offsets are scanned on the device (no reference words down / offsets up round trip)
// and leave as u32, the positions as u16 (15 bits + the strand in bit 15): 10 bytes per tuple + 5 per read over the link instead of
// 12 + 17.  For streaming callers (pipeline.cpp); a group of 16 lanes per sequence.
namespace {
__global__ void k_gather_narrow(const u64 *hash, const u32 *pos, const u64 *refs, const u64 *wfirst, const u64 *wcount, const u64 *dstoff, u64 n,
                                u64 *ohash, u16 *opos, u32 *ooff, u32 *flag) {
    const u64 grp = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 4, ng = ((u64)gridDim.x * blockDim.x) >> 4;
    const u32 gl = threadIdx.x & 15;
    u32 wide = 0;
    for (u64 r = grp; r <= n; r += ng) {
        if (gl == 0) ooff[r] = (u32)dstoff[r];
        if (r == n) break;
        u64 b, cnt, st;
        seq_span(refs, wfirst, wcount, r, b, cnt, st);
        const u64 d = dstoff[r];
        for (u64 t = gl; t < cnt; t += 16) {
            ohash[d + t] = hash[b + t * st];
            if (opos) {
                const u32 pv = pos[b + t * st];
                wide |= pv & 0x7fff8000u;
                opos[d + t] = (u16)((pv & 0x7fffu) | ((pv >> 16) & 0x8000u));
            }
        }
    }
    if (wide) atomicOr(flag, 1u);
}
}  // namespace

extern "C" int bsk_result_fetch_narrow(bsk_ctx *ctx, const bsk_result *r, uint64_t first, uint64_t count, uint32_t *offsets, uint8_t *status,
                                       uint64_t *hash, uint16_t *pos, uint64_t tuple_cap, uint64_t *n_tuples) {
    if (!ctx || !r || !offsets || !hash) return fail_arg(ctx, "bsk_result_fetch_narrow: null argument");
    if (r->ctx != ctx) return fail_arg(ctx, "bsk_result_fetch_narrow: result belongs to another context");
    if (first + count > r->n) return fail_arg(ctx, "bsk_result_fetch_narrow: range outside result");
    if (pos && !r->pos) return fail_arg(ctx, "bsk_result_fetch_narrow: this kind has implicit positions");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    auto pool = [&](int slot, size_t bytes, void **outp) -> hipError_t {
        if (ctx->tmp_cap[slot] < bytes) {
            (void)hipFree(ctx->tmp[slot]);
            ctx->tmp[slot] = nullptr;
            ctx->tmp_cap[slot] = 0;
            const size_t want = bytes + bytes / 4 + 256;
            const hipError_t e = hipMalloc(&ctx->tmp[slot], want);
            if (e != hipSuccess) return e;
            ctx->tmp_cap[slot] = want;
        }
        *outp = ctx->tmp[slot];
        return hipSuccess;
    };
    u64 *offs = nullptr, *part = nullptr, *oh = nullptr;
    u16 *op = nullptr;
    u32 *oo = nullptr;
    HIPCHK(ctx, pool(12, (count + 2) * 8 + (count + 2) * 4, (void **)&offs));
    oo = reinterpret_cast<u32 *>(offs + count + 2);
    HIPCHK(ctx, pool(17, ((count + SCAN_CHUNK - 1) / SCAN_CHUNK + 2) * 8, (void **)&part));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_total, 0, 32, st));
    HIPCHK(ctx, scan_counts(st, CountOf{r->refs ? r->refs + first : nullptr, r->refs ? nullptr : r->wfirst + first, r->refs ? nullptr : r->wcount + first}, count, part, offs,
                            ctx->d_total + 1, (u64 *)nullptr));
    u64 T = r->n_tuples;
    if (first != 0 || count != r->n) {  // (the whole result's count is known on the host: no round trip)
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_pinned, ctx->d_total + 1, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        T = ctx->h_pinned[0];
    }
    if (n_tuples) *n_tuples = T;
    if (T > tuple_cap) return fail_arg(ctx, "bsk_result_fetch_narrow: tuple_cap too small");
    if (T >= (1ULL << 32)) return fail_arg(ctx, "bsk_result_fetch_narrow: 2^32 tuples or more (fetch a smaller range)");
    HIPCHK(ctx, pool(13, (T + 1) * 8, (void **)&oh));
    if (pos) HIPCHK(ctx, pool(14, (T + 1) * 2, (void **)&op));
    if (r->refs && !ctx->opt.no_group_gather && r->unit_rows)  // unit rows
        hipLaunchKernelGGL((k_gather_groups<true, true>), dim3(grid_of(ctx, count + 64, 1)), dim3(64), 0, st, r->hash, pos ? r->pos : nullptr, r->refs + first, offs,
                           count, oh, (void *)op, oo, reinterpret_cast<u32 *>(ctx->d_total + 3));
    else if (r->refs && !ctx->opt.no_group_gather)
        hipLaunchKernelGGL((k_gather_groups<true, false>), dim3(grid_of(ctx, count + 64, 256)), dim3(256), 0, st, r->hash, pos ? r->pos : nullptr, r->refs + first, offs,
                           count, oh, (void *)op, oo, reinterpret_cast<u32 *>(ctx->d_total + 3));
    else
        hipLaunchKernelGGL(k_gather_narrow, dim3(grid_of(ctx, (count + 1) * 16, 256)), dim3(256), 0, st, r->hash, pos ? r->pos : nullptr,
                           r->refs ? r->refs + first : nullptr, r->refs ? nullptr : r->wfirst + first, r->refs ? nullptr : r->wcount + first, offs, count, oh, op, oo,
                           reinterpret_cast<u32 *>(ctx->d_total + 3));
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(offsets, oo, (count + 1) * 4, hipMemcpyDeviceToHost, st));
    if (status && count) HIPCHK(ctx, hipMemcpyAsync(status, r->status + first, count, hipMemcpyDeviceToHost, st));
    if (T) HIPCHK(ctx, hipMemcpyAsync(hash, oh, T * 8, hipMemcpyDeviceToHost, st));
    if (pos && T) HIPCHK(ctx, hipMemcpyAsync(pos, op, T * 2, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_pinned + 1, ctx->d_total + 3, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    if (pos && (ctx->h_pinned[1] & 1)) {
        ctx->err = "bsk_result_fetch_narrow: a position does not fit 15 bits (reads of 32 768 bases or more: use bsk_result_fetch)";
        return BSK_ERR_UNSUPPORTED;
    }
    return BSK_OK;
}

extern "C" void bsk_sets_release(bsk_sets *s) {
    if (!s) return;
    if (s->ctx) (void)hipSetDevice(s->ctx->device);
    (void)hipFree(s->offsets);
    (void)hipFree(s->values);
    delete s;
}

static int sets_impl(bsk_ctx *ctx, const bsk_result *r, int scope, int scale, bsk_sets *reuse, bsk_sets **out);
extern "C" int bsk_result_sets(bsk_ctx *ctx, const bsk_result *r, int scope, int scale, bsk_sets **out) {
    return sets_impl(ctx, r, scope, scale, nullptr, out);
}
// The same into an EXISTING sets object (*sets may be NULL the first time): its device arrays are kept and only grow, so a streaming
// caller -- one chunk after the other through one object per stream -- allocates nothing in steady state (hipMalloc / hipFree
// synchronise the whole device and would serialise the streams; bsk_batch_refill_ascii is the same idea for batches).  On error the
// object is released and *sets is NULL.
extern "C" int bsk_result_sets_reuse(bsk_ctx *ctx, const bsk_result *r, int scope, int scale, bsk_sets **sets) {
    if (!sets) return fail_arg(ctx, "bsk_result_sets_reuse: null argument");
    bsk_sets *old = *sets;
    if (old && old->ctx != ctx) return fail_arg(ctx, "bsk_result_sets_reuse: the sets belong to another context");
    *sets = nullptr;
    return sets_impl(ctx, r, scope, scale, old, sets);
}
static int sets_impl(bsk_ctx *ctx, const bsk_result *r, int scope, int scale, bsk_sets *reuse, bsk_sets **out) {
    if (!ctx || !r || !out) {
        if (reuse) bsk_sets_release(reuse);
        return fail_arg(ctx, "bsk_result_sets: null argument");
    }
    *out = nullptr;
    if (r->ctx != ctx || (scope != BSK_SETS_PER_SEQUENCE && scope != BSK_SETS_WHOLE_BATCH) || scale < 0) {
        if (reuse) bsk_sets_release(reuse);
        return fail_arg(ctx, r->ctx != ctx ? "bsk_result_sets: result belongs to another context" : scale < 0 ? "bsk_result_sets: bad scale" : "bsk_result_sets: bad scope");
    }
    if (hipSetDevice(ctx->device) != hipSuccess) {
        if (reuse) bsk_sets_release(reuse);
        return fail_arg(ctx, "bsk_result_sets: hipSetDevice");
    }
    const u64 n = r->n;
    const u64 maxhash = scale > 1 ? ~0ULL / (u64)scale : ~0ULL;
    const u64 n_sets = scope == BSK_SETS_WHOLE_BATCH ? 1 : n;
    hipStream_t st = ctx->stream;
    u64 *offs = nullptr /* pooled */, *vin = nullptr, *vsorted = nullptr, *pos = nullptr;
    u8 *head = nullptr;
    u32 *keep = nullptr;
    void *tmp = nullptr;
    bsk_sets *res = reuse;
    auto done = [&](int code) {
        if (code != BSK_OK && res) bsk_sets_release(res);
        return code;
    };
    auto grow = [&](u64 **p, size_t *cap, size_t bytes) -> hipError_t {  // the sets' own arrays: grow-only
        if (*cap >= bytes && *p) return hipSuccess;
        (void)hipFree(*p);
        *p = nullptr;
        *cap = 0;
        const size_t want = reuse ? bytes + bytes / 4 + 256 : bytes;
        const hipError_t e = hipMalloc(p, want);
        if (e == hipSuccess) *cap = want;
        return e;
    };
#define SCHK(call)                                                  \
    do {                                                            \
        hipError_t e__ = (call);                                    \
        if (e__ != hipSuccess) return done(fail_hip(ctx, e__, #call)); \
    } while (0)
    // temporaries of every call come from the context's grow-only pool (hipMalloc / hipFree synchronise the device)
    auto pool = [&](int slot, size_t bytes, void **outp) -> hipError_t {
        if (ctx->tmp_cap[slot] < bytes) {
            (void)hipFree(ctx->tmp[slot]);
            ctx->tmp[slot] = nullptr;
            ctx->tmp_cap[slot] = 0;
            const size_t want = bytes + bytes / 4 + 256;
            const hipError_t e = hipMalloc(&ctx->tmp[slot], want);
            if (e != hipSuccess) return e;
            ctx->tmp_cap[slot] = want;
        }
        *outp = ctx->tmp[slot];
        return hipSuccess;
    };
    // 1. where every sequence's values go in a dense copy: exclusive scan of the counts (+ the largest count)
    u64 *part = nullptr;
    SCHK(pool(0, (n + 2) * 8, (void **)&offs));
    SCHK(pool(1, ((n + SCAN_CHUNK - 1) / SCAN_CHUNK + 2) * 8, (void **)&part));
    SCHK(hipMemsetAsync(ctx->d_total, 0, 16, st));  // [0] largest count, [1] total
    SCHK(scan_counts(st, CountOf{r->refs, r->wfirst, r->wcount}, n, part, offs, ctx->d_total + 1, ctx->d_total));
    u64 N = 0, max_count = 0;
    SCHK(hipMemcpyAsync(&N, ctx->d_total + 1, 8, hipMemcpyDeviceToHost, st));
    SCHK(hipMemcpyAsync(&max_count, ctx->d_total, 8, hipMemcpyDeviceToHost, st));
    SCHK(hipStreamSynchronize(st));
    size_t tb = 0;
    if (N >= (1ULL << 32)) {
        ctx->err = "bsk_result_sets: more than 2^32 values in one call (split the batch)";
        return done(BSK_ERR_UNSUPPORTED);
    }
    if (!res) res = new (std::nothrow) bsk_sets();
    if (!res) return done(BSK_ERR_NOMEM);
    res->ctx = ctx;
    res->n_sets = n_sets;
    res->n_values = 0;
    SCHK(grow(&res->offsets, &res->c_offsets, (n_sets + 1) * 8));
    SCHK(grow(&res->values, &res->c_values, (N ? N : 1) * 8));
    if (N == 0) {
        SCHK(hipMemsetAsync(res->offsets, 0, (n_sets + 1) * 8, st));
        SCHK(hipStreamSynchronize(st));
        *out = res;
        return done(BSK_OK);
    }
    if (scope == BSK_SETS_PER_SEQUENCE && max_count <= SMALL_CAP && !ctx->opt.sets_no_small) {
        // short reads: one sequence per row of 16 lanes, bitonic network over DPP moves (k_sets_rows)
        u64 *ucount = nullptr, *dense = nullptr;
        SCHK(pool(2, N * 8, (void **)&dense));          // the sequences' distinct values at their input offsets
        SCHK(pool(3, (n + 2) * 8, (void **)&ucount));
        const unsigned sgrid = (unsigned)std::max<u64>(1, std::min<u64>((n + 15) / 16, (u64)ctx->cus * 32));
        hipLaunchKernelGGL(k_sets_rows, dim3(sgrid), dim3(256), 0, st, r->hash, r->refs, r->wfirst, r->wcount, offs, n, maxhash, dense, ucount);
        SCHK(hipGetLastError());
        SCHK(scan_counts(st, ArrayOf{ucount}, n, part, res->offsets, ctx->d_total + 1, (u64 *)nullptr));
        if (!ctx->opt.no_group_gather)
            hipLaunchKernelGGL(k_move_groups, dim3((unsigned)std::max<u64>(1, std::min<u64>((n + 255) / 256, (u64)ctx->cus * 16))), dim3(256), 0, st, dense, offs,
                               res->offsets, ucount, n, res->values);
        else
            hipLaunchKernelGGL(k_move_seqs, dim3((unsigned)std::max<u64>(1, std::min<u64>((n + 7) / 8, (u64)ctx->cus * 32))), dim3(256), 0, st, dense, offs, res->offsets,
                               ucount, n, res->values);
        SCHK(hipGetLastError());
        u64 M = 0;
        SCHK(hipMemcpyAsync(&M, ctx->d_total + 1, 8, hipMemcpyDeviceToHost, st));
        SCHK(hipStreamSynchronize(st));
        res->n_values = M;
        *out = res;
        return done(BSK_OK);
    }
    SCHK(pool(2, N * 8, (void **)&vin));  // (all temporaries are pooled: in a process that has already cycled ~100 GB of batches a
    SCHK(pool(5, N * 8, (void **)&vsorted));  //  hipMalloc / hipFree of these buffers cost 70-200 ms per call against 10-18 ms of kernels)
    hipLaunchKernelGGL(k_gather_values, dim3(grid_of(ctx, n * 16, 256)), dim3(256), 0, st, r->hash, r->refs, r->wfirst, r->wcount, offs, n,
                       maxhash, vin);
    SCHK(hipGetLastError());
    // 2. sort inside every set
    u64 *set_offs = offs;  // per-sequence scope: the gather offsets ARE the segment offsets
    if (scope == BSK_SETS_WHOLE_BATCH) {
        const u64 two[2] = {0, N};
        SCHK(hipStreamSynchronize(st));                           // the gather above still reads offs
        SCHK(hipMemcpy(offs, two, 16, hipMemcpyHostToDevice));   // offs has n + 2 >= 2 entries
        SCHK(rocprim::radix_sort_keys(nullptr, tb, vin, vsorted, (size_t)N, 0, 64, st));
        SCHK(pool(9, tb ? tb : 8, &tmp));
        SCHK(rocprim::radix_sort_keys(tmp, tb, vin, vsorted, (size_t)N, 0, 64, st));
    } else {
        SCHK(rocprim::segmented_radix_sort_keys(nullptr, tb, vin, vsorted, (unsigned)N, (unsigned)n, offs, offs + 1, 0, 64, st));
        SCHK(pool(9, tb ? tb : 8, &tmp));
        SCHK(rocprim::segmented_radix_sort_keys(tmp, tb, vin, vsorted, (unsigned)N, (unsigned)n, offs, offs + 1, 0, 64, st));
    }
    // 3. first occurrences that pass the filter -> compacted sets
    SCHK(pool(6, N, (void **)&head));
    SCHK(pool(7, N * 4, (void **)&keep));
    SCHK(pool(8, (N + 1) * 8, (void **)&pos));
    SCHK(hipMemsetAsync(head, 0, N, st));
    hipLaunchKernelGGL(k_mark_heads, dim3(grid_of(ctx, n_sets, 256)), dim3(256), 0, st, set_offs, n_sets, head);
    hipLaunchKernelGGL(k_flag_unique, dim3(grid_of(ctx, N, 256)), dim3(256), 0, st, vsorted, head, N, maxhash, keep);
    SCHK(hipGetLastError());
    // (the library's own three-kernel scan: rocprim::exclusive_scan over a transform iterator was 1.1 MB of the shared object)
    u64 *part2 = nullptr;
    SCHK(pool(9, ((N + SCAN_CHUNK - 1) / SCAN_CHUNK + 2) * 8, (void **)&part2));
    SCHK(scan_counts(st, KeepOf{keep}, N, part2, pos, ctx->d_total + 1, (u64 *)nullptr));
    u64 M = 0;
    SCHK(hipMemcpyAsync(&M, ctx->d_total + 1, 8, hipMemcpyDeviceToHost, st));
    SCHK(hipStreamSynchronize(st));
    hipLaunchKernelGGL(k_scatter_unique, dim3(grid_of(ctx, N, 256)), dim3(256), 0, st, vsorted, keep, pos, N, res->values);
    hipLaunchKernelGGL(k_new_offsets, dim3(grid_of(ctx, n_sets + 1, 256)), dim3(256), 0, st, set_offs, pos, n_sets, N, M, res->offsets);
    SCHK(hipGetLastError());
    SCHK(hipStreamSynchronize(st));
    res->n_values = M;
#undef SCHK
    *out = res;
    return done(BSK_OK);
}

extern "C" int bsk_sets_info(const bsk_sets *s, uint64_t *n_sets, uint64_t *n_values) {
    if (!s) return BSK_ERR_ARG;
    if (n_sets) *n_sets = s->n_sets;
    if (n_values) *n_values = s->n_values;
    return BSK_OK;
}

extern "C" int bsk_sets_device(const bsk_sets *s, const uint64_t **offsets, const uint64_t **values) {
    if (!s) return BSK_ERR_ARG;
    if (offsets) *offsets = (const uint64_t *)s->offsets;
    if (values) *values = (const uint64_t *)s->values;
    return BSK_OK;
}

namespace {
__global__ __launch_bounds__(256) void k_offsets_u32(const u64 *in, u64 n1, u32 *out) {
    for (u64 i = blockIdx.x * 256ull + threadIdx.x; i < n1; i += (u64)gridDim.x * 256ull) out[i] = (u32)in[i];
}
}  // namespace
// All sets of `s` to the host in the narrow form of bsk_result_fetch_narrow: u32 offsets[n_sets + 1] (narrowed on the device; a call
// holds fewer than 2^32 values) and the values, both copied on the context's stream straight into the caller's buffers -- pinned ones
// make the copies asynchronous to the other streams of a pipeline.  4 bytes per set + 8 per value over the link.
extern "C" int bsk_sets_fetch_narrow(bsk_ctx *ctx, const bsk_sets *s, uint32_t *offsets, uint64_t *values, uint64_t value_cap) {
    if (!ctx || !s || !offsets) return fail_arg(ctx, "bsk_sets_fetch_narrow: null argument");
    if (s->ctx != ctx) return fail_arg(ctx, "bsk_sets_fetch_narrow: the sets belong to another context");
    if (values && s->n_values > value_cap) return fail_arg(ctx, "bsk_sets_fetch_narrow: value_cap too small");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t need = (s->n_sets + 1) * 4;
    if (ctx->tmp_cap[20] < need) {
        (void)hipFree(ctx->tmp[20]);
        ctx->tmp[20] = nullptr;
        ctx->tmp_cap[20] = 0;
        HIPCHK(ctx, hipMalloc(&ctx->tmp[20], need + need / 4 + 256));
        ctx->tmp_cap[20] = need + need / 4 + 256;
    }
    u32 *o32 = (u32 *)ctx->tmp[20];
    hipLaunchKernelGGL(k_offsets_u32, dim3((unsigned)std::max<u64>(1, std::min<u64>((s->n_sets + 256) / 256, (u64)ctx->cus * 8))), dim3(256), 0, ctx->stream, s->offsets,
                       s->n_sets + 1, o32);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(offsets, o32, need, hipMemcpyDeviceToHost, ctx->stream));
    if (values && s->n_values) HIPCHK(ctx, hipMemcpyAsync(values, s->values, s->n_values * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return BSK_OK;
}

extern "C" int bsk_sets_fetch(bsk_ctx *ctx, const bsk_sets *s, uint64_t first, uint64_t count, uint64_t *offsets, uint64_t *values,
                              uint64_t value_cap) {
    if (!ctx || !s || !offsets) return fail_arg(ctx, "bsk_sets_fetch: null argument");
    if (first + count > s->n_sets) return fail_arg(ctx, "bsk_sets_fetch: range outside the sets");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::vector<u64> o(count + 1);
    HIPCHK(ctx, hipMemcpy(o.data(), s->offsets + first, (count + 1) * 8, hipMemcpyDeviceToHost));
    const u64 nv = o[count] - o[0];
    for (u64 i = 0; i <= count; ++i) offsets[i] = o[i] - o[0];
    if (!values) return BSK_OK;
    if (nv > value_cap) return fail_arg(ctx, "bsk_sets_fetch: value_cap too small");
    if (nv) HIPCHK(ctx, hipMemcpy(values, s->values + o[0], nv * 8, hipMemcpyDeviceToHost));
    return BSK_OK;
}
