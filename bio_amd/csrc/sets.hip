// sets.hip -- sketch SETS: the distinct hash values of a result, ascending, per sequence or for the whole batch,
// optionally FracMinHash-filtered (hash <= MaxUint64/scale, the rule of iterator.go:181-185).  This is the form the
// downstream consumers of the reference's sketches keep on disk (kmcp / unikmer: sorted unique uint64 lists):
// collecting Next() values into a slice, sort, de-duplicate.  Done on the device with rocPRIM (segmented radix sort,
// scan) so that only the final sets cross PCIe.
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
};

namespace {

// where the values of sequence r are: packed reference word or the wide arrays (host_types.hpp)
__device__ __forceinline__ void seq_span(const u64 *refs, const u64 *wfirst, const u64 *wcount, u64 r, u64 &first, u64 &cnt) {
    if (refs) {
        first = refs[r] >> 24;
        cnt = refs[r] & 0xffffffULL;
    } else {
        first = wfirst[r];
        cnt = wcount[r];
    }
}

__global__ void k_counts(const u64 *refs, const u64 *wfirst, const u64 *wcount, u64 n, u64 *cnt) {
    for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (u64)gridDim.x * blockDim.x) {
        u64 f, c;
        seq_span(refs, wfirst, wcount, r, f, c);
        cnt[r] = c;
    }
}

// dense copy of every sequence's values (one wavefront per sequence); values above maxhash become the sentinel ~0 so that
// they sort to the end of their segment
__global__ void k_gather_values(const u64 *hash, const u64 *refs, const u64 *wfirst, const u64 *wcount, const u64 *dst, u64 n, u64 maxhash,
                                u64 *out) {
    const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 r = wave; r < n; r += nw) {
        u64 f, c;
        seq_span(refs, wfirst, wcount, r, f, c);
        const u64 d = dst[r];
        for (u64 t = threadIdx.x & 63; t < c; t += 64) {
            const u64 h = hash[f + t];
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
// A segmented radix sort spends eight digit passes on segments of two dozen keys.  Here a sequence is sorted by a GROUP of
// G = 32 (or 64) lanes holding one value each: the group loads the sequence's values with one coalesced load, runs a bitonic
// network across its lanes (log2(G)(log2(G)+1)/2 exchange steps: ds_bpermute + 64-bit compare + select, no LDS memory, a
// dozen registers -- full occupancy), drops duplicates by comparing with the lane below, and stores the distinct values as
// one dense run at the sequence's input offset (ballot + mbcnt).  k_move_seqs then shifts the runs to their final offsets.
#define SMALL_CAP 64
template <int G>
__global__ __launch_bounds__(256) void k_sets_small(const u64 *hash, const u64 *refs, const u64 *wfirst, const u64 *wcount, const u64 *offs, u64 n,
                                                    u64 maxhash, u64 *tmp, u64 *ucount) {
    constexpr int PER_WAVE = 64 / G;
    const int lane = threadIdx.x & 63, gl = lane & (G - 1), grp = lane / G;
    const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((u64)gridDim.x * blockDim.x) >> 6;
    const u64 ngroups = (n + PER_WAVE - 1) / PER_WAVE;  // wave iterations
    for (u64 it = wave; it < ngroups; it += nw) {
        const u64 r = it * PER_WAVE + grp;
        u64 first = 0, c = 0;
        if (r < n) seq_span(refs, wfirst, wcount, r, first, c);
        u64 v = ~0ULL;
        bool valid = false;
        if ((u64)gl < c) {
            v = hash[first + gl];
            valid = v <= maxhash;
            if (!valid) v = ~0ULL;  // filtered values sort to the end with the padding
        }
        const u64 vm = __ballot(valid);
        const u32 nvalid = (u32)__builtin_popcountll(G == 64 ? vm : ((vm >> (grp * G)) & ((1ULL << (G & 63)) - 1)));
#pragma unroll
        for (int k = 2; k <= G; k <<= 1) {
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                const u64 o = __shfl_xor(v, j, 64);
                const bool up = (gl & k) == 0;             // this block of k lanes sorts ascending
                const bool lower = (gl & j) == 0;          // this lane keeps the smaller one when ascending
                const bool take_min = lower == up;
                const bool o_less = o < v;
                v = (take_min == o_less) ? o : v;
            }
        }
        // ascending within the group; lanes < nvalid hold the values that passed the filter
        const u64 below = __shfl_up(v, 1, 64);
        const bool keep = (u32)gl < nvalid && (gl == 0 || v != below);
        const u64 km = __ballot(keep);
        const u64 gm = G == 64 ? km : ((km >> (grp * G)) & ((1ULL << (G & 63)) - 1));
        const u32 rank = (u32)__builtin_popcountll(gm & ((1ULL << gl) - 1));
        if (keep) tmp[offs[r] + rank] = v;
        if (gl == 0 && r < n) ucount[r] = (u64)__builtin_popcountll(gm);
    }
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
__global__ void k_max_count(const u64 *cnt, u64 n, u64 *mx) {
    u64 m = 0;
    for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (u64)gridDim.x * blockDim.x) m = cnt[r] > m ? cnt[r] : m;
    for (int d = 32; d; d >>= 1) {
        const u64 t = __shfl_xor(m, d, 64);
        m = t > m ? t : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(mx, m);
}

int grid_of(bsk_ctx *ctx, u64 items, int block) {
    const u64 g = (items + block - 1) / block;
    return (int)std::max<u64>(1, std::min<u64>(g, (u64)ctx->cus * 16));
}

}  // namespace

extern "C" void bsk_sets_release(bsk_sets *s) {
    if (!s) return;
    if (s->ctx) (void)hipSetDevice(s->ctx->device);
    (void)hipFree(s->offsets);
    (void)hipFree(s->values);
    delete s;
}

extern "C" int bsk_result_sets(bsk_ctx *ctx, const bsk_result *r, int scope, int scale, bsk_sets **out) {
    if (!ctx || !r || !out) return fail_arg(ctx, "bsk_result_sets: null argument");
    if (r->ctx != ctx) return fail_arg(ctx, "bsk_result_sets: result belongs to another context");
    if (scope != BSK_SETS_PER_SEQUENCE && scope != BSK_SETS_WHOLE_BATCH) return fail_arg(ctx, "bsk_result_sets: bad scope");
    if (scale < 0) return fail_arg(ctx, "bsk_result_sets: bad scale");
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const u64 n = r->n;
    const u64 maxhash = scale > 1 ? ~0ULL / (u64)scale : ~0ULL;
    const u64 n_sets = scope == BSK_SETS_WHOLE_BATCH ? 1 : n;
    hipStream_t st = ctx->stream;
    u64 *cnt = nullptr, *offs = nullptr, *vin = nullptr, *vsorted = nullptr, *pos = nullptr;
    u8 *head = nullptr;
    u32 *keep = nullptr;
    void *tmp = nullptr;
    bsk_sets *res = nullptr;
    auto done = [&](int code) {
        (void)hipFree(cnt);
        (void)hipFree(offs);
        (void)hipFree(vin);
        (void)hipFree(vsorted);
        (void)hipFree(pos);
        (void)hipFree(head);
        (void)hipFree(keep);
        (void)hipFree(tmp);
        if (code != BSK_OK && res) bsk_sets_release(res);
        return code;
    };
#define SCHK(call)                                                  \
    do {                                                            \
        hipError_t e__ = (call);                                    \
        if (e__ != hipSuccess) return done(fail_hip(ctx, e__, #call)); \
    } while (0)
    // 1. dense copy of the values, sequence after sequence
    SCHK(hipMalloc(&cnt, (n + 66) * 8));
    SCHK(hipMalloc(&offs, (n + 66) * 8));  // (+64: the small-set path reads offs[64 * unit] of a last, partial unit)
    SCHK(hipMemsetAsync(cnt, 0, (n + 66) * 8, st));
    SCHK(hipMemsetAsync(ctx->d_total, 0, 8, st));
    if (n) {
        hipLaunchKernelGGL(k_counts, dim3(grid_of(ctx, n, 256)), dim3(256), 0, st, r->refs, r->wfirst, r->wcount, n, cnt);
        hipLaunchKernelGGL(k_max_count, dim3(grid_of(ctx, n, 256)), dim3(256), 0, st, cnt, n, ctx->d_total);
    }
    size_t tb = 0;
    SCHK(rocprim::exclusive_scan(nullptr, tb, cnt, offs, (u64)0, n + 65, rocprim::plus<u64>(), st));
    SCHK(hipMalloc(&tmp, tb ? tb : 8));
    SCHK(rocprim::exclusive_scan(tmp, tb, cnt, offs, (u64)0, n + 65, rocprim::plus<u64>(), st));
    u64 N = 0, max_count = 0;
    SCHK(hipMemcpyAsync(&N, offs + n, 8, hipMemcpyDeviceToHost, st));
    SCHK(hipMemcpyAsync(&max_count, ctx->d_total, 8, hipMemcpyDeviceToHost, st));
    SCHK(hipStreamSynchronize(st));
    (void)hipFree(tmp);
    tmp = nullptr;
    if (N >= (1ULL << 32)) {
        ctx->err = "bsk_result_sets: more than 2^32 values in one call (split the batch)";
        return done(BSK_ERR_UNSUPPORTED);
    }
    res = new (std::nothrow) bsk_sets();
    if (!res) return done(BSK_ERR_NOMEM);
    res->ctx = ctx;
    res->n_sets = n_sets;
    SCHK(hipMalloc(&res->offsets, (n_sets + 1) * 8));
    SCHK(hipMalloc(&res->values, (N ? N : 1) * 8));
    if (N == 0) {
        SCHK(hipMemsetAsync(res->offsets, 0, (n_sets + 1) * 8, st));
        SCHK(hipStreamSynchronize(st));
        *out = res;
        return done(BSK_OK);
    }
    if (scope == BSK_SETS_PER_SEQUENCE && max_count <= SMALL_CAP && !getenv("BSK_SETS_NO_SMALL")) {
        // short reads: one sequence per group of 32 / 64 lanes, bitonic network across the lanes (k_sets_small)
        u64 *ucount = nullptr, *ooffs = nullptr;
        SCHK(hipMalloc(&vin, N * 8));                    // the sequences' distinct values at their input offsets
        SCHK(hipMalloc(&keep, (n + 66) * 8));            // ucount (as u64)
        SCHK(hipMalloc(&vsorted, (n + 66) * 8));         // output offsets
        ucount = (u64 *)keep;
        ooffs = vsorted;
        SCHK(hipMemsetAsync(ucount, 0, (n + 66) * 8, st));
        const unsigned sgrid = (unsigned)std::max<u64>(1, std::min<u64>((n + 7) / 8, (u64)ctx->cus * 32));
        if (max_count <= 32)
            hipLaunchKernelGGL(k_sets_small<32>, dim3(sgrid), dim3(256), 0, st, r->hash, r->refs, r->wfirst, r->wcount, offs, n, maxhash, vin, ucount);
        else
            hipLaunchKernelGGL(k_sets_small<64>, dim3(sgrid), dim3(256), 0, st, r->hash, r->refs, r->wfirst, r->wcount, offs, n, maxhash, vin, ucount);
        SCHK(hipGetLastError());
        SCHK(rocprim::exclusive_scan(nullptr, tb, ucount, ooffs, (u64)0, n + 1, rocprim::plus<u64>(), st));
        SCHK(hipMalloc(&tmp, tb ? tb : 8));
        SCHK(rocprim::exclusive_scan(tmp, tb, ucount, ooffs, (u64)0, n + 1, rocprim::plus<u64>(), st));
        hipLaunchKernelGGL(k_move_seqs, dim3((unsigned)std::max<u64>(1, std::min<u64>((n + 7) / 8, (u64)ctx->cus * 32))), dim3(256), 0, st, vin, offs, ooffs, ucount, n,
                           res->values);
        SCHK(hipGetLastError());
        SCHK(hipMemcpyAsync(res->offsets, ooffs, (n + 1) * 8, hipMemcpyDeviceToDevice, st));
        u64 M = 0;
        SCHK(hipMemcpyAsync(&M, ooffs + n, 8, hipMemcpyDeviceToHost, st));
        SCHK(hipStreamSynchronize(st));
        res->n_values = M;
        *out = res;
        return done(BSK_OK);
    }
    SCHK(hipMalloc(&vin, N * 8));
    SCHK(hipMalloc(&vsorted, N * 8));
    hipLaunchKernelGGL(k_gather_values, dim3(grid_of(ctx, n * 64, 256)), dim3(256), 0, st, r->hash, r->refs, r->wfirst, r->wcount, offs, n,
                       maxhash, vin);
    SCHK(hipGetLastError());
    // 2. sort inside every set
    u64 *set_offs = offs;  // per-sequence scope: the gather offsets ARE the segment offsets
    if (scope == BSK_SETS_WHOLE_BATCH) {
        const u64 two[2] = {0, N};
        SCHK(hipStreamSynchronize(st));                           // the gather above still reads offs
        SCHK(hipMemcpy(offs, two, 16, hipMemcpyHostToDevice));   // offs has n + 2 >= 2 entries
        SCHK(rocprim::radix_sort_keys(nullptr, tb, vin, vsorted, (size_t)N, 0, 64, st));
        SCHK(hipMalloc(&tmp, tb ? tb : 8));
        SCHK(rocprim::radix_sort_keys(tmp, tb, vin, vsorted, (size_t)N, 0, 64, st));
    } else {
        SCHK(rocprim::segmented_radix_sort_keys(nullptr, tb, vin, vsorted, (unsigned)N, (unsigned)n, offs, offs + 1, 0, 64, st));
        SCHK(hipMalloc(&tmp, tb ? tb : 8));
        SCHK(rocprim::segmented_radix_sort_keys(tmp, tb, vin, vsorted, (unsigned)N, (unsigned)n, offs, offs + 1, 0, 64, st));
    }
    (void)hipFree(tmp);
    tmp = nullptr;
    // 3. first occurrences that pass the filter -> compacted sets
    SCHK(hipMalloc(&head, N));
    SCHK(hipMalloc(&keep, N * 4));
    SCHK(hipMalloc(&pos, (N + 1) * 8));
    SCHK(hipMemsetAsync(head, 0, N, st));
    hipLaunchKernelGGL(k_mark_heads, dim3(grid_of(ctx, n_sets, 256)), dim3(256), 0, st, set_offs, n_sets, head);
    hipLaunchKernelGGL(k_flag_unique, dim3(grid_of(ctx, N, 256)), dim3(256), 0, st, vsorted, head, N, maxhash, keep);
    SCHK(hipGetLastError());
    auto keep64 = rocprim::make_transform_iterator(keep, [] __device__(u32 k) -> u64 { return (u64)k; });
    SCHK(rocprim::exclusive_scan(nullptr, tb, keep64, pos, (u64)0, (size_t)N, rocprim::plus<u64>(), st));
    SCHK(hipMalloc(&tmp, tb ? tb : 8));
    SCHK(rocprim::exclusive_scan(tmp, tb, keep64, pos, (u64)0, (size_t)N, rocprim::plus<u64>(), st));
    u64 last_pos = 0;
    u32 last_keep = 0;
    SCHK(hipMemcpyAsync(&last_pos, pos + (N - 1), 8, hipMemcpyDeviceToHost, st));
    SCHK(hipMemcpyAsync(&last_keep, keep + (N - 1), 4, hipMemcpyDeviceToHost, st));
    SCHK(hipStreamSynchronize(st));
    const u64 M = last_pos + last_keep;
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
