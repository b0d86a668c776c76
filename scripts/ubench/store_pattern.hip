// dev microbenchmark: what write bandwidth does the sketch kernels' OUTPUT PATTERN reach on its own (no hashing)?
// 3072 persistent wavefronts (12 per CU, as k_minimizer_ring) each own units (slabs of ROWS x 64 tuples: hash u64 + pos u32) and
// write them the way the kernels do:
//   rows     row by row: 512 B of hashes + 256 B of positions per row (k_minimizer_ring), DELAY "hash" cycles between groups of 4 rows
//   dense    the unit's tuples as one contiguous run, 512 B + 256 B per store pair (k_minimizer_pk's copy-out)
//   x4       rows, but 16-byte stores (two rows per instruction are not contiguous, so: lanes 0..31 one row, 32..63 the next)
// hipcc --offload-arch=gfx950 -O2 -o store_pattern store_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;

template <int MODE, bool NT>
__global__ __launch_bounds__(64) void k_pat(u64 *hash, u32 *pos, u32 nunits, u32 rows, u32 fill, u32 delay) {
    const u32 lane = threadIdx.x;
    for (u32 unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
        const u64 base = (u64)unit * rows * 64;
        u64 *gh = hash + base;
        u32 *gp = pos + base;
        if (MODE == 0) {
            for (u32 t = 0; t < fill; t += 4) {
                if (delay) __builtin_amdgcn_s_sleep(0), __builtin_amdgcn_s_sleep(0);
                for (u32 d = 0; d < delay; ++d) __builtin_amdgcn_s_sleep(16);  // ~1024 cycles per unit of delay
#pragma unroll
                for (u32 j = 0; j < 4; ++j) {
                    if (NT) {
                        __builtin_nontemporal_store((u64)unit + t, gh + (t + j) * 64 + lane);
                        __builtin_nontemporal_store(t, gp + (t + j) * 64 + lane);
                    } else {
                        gh[(t + j) * 64 + lane] = (u64)unit + t;
                        gp[(t + j) * 64 + lane] = t;
                    }
                }
            }
        } else if (MODE == 1) {  // dense: fill * 64 tuples contiguous
            for (u32 d = 0; d < delay * (fill / 4); ++d) __builtin_amdgcn_s_sleep(16);
            for (u32 t = 0; t < fill; ++t) {
                if (NT) {
                    __builtin_nontemporal_store((u64)unit + t, gh + t * 64 + lane);
                    __builtin_nontemporal_store(t, gp + t * 64 + lane);
                } else {
                    gh[t * 64 + lane] = (u64)unit + t;
                    gp[t * 64 + lane] = t;
                }
            }
        }
    }
}

template <int MODE, bool NT>
void run(const char *name, u64 *hash, u32 *pos, u32 nunits, u32 rows, u32 fill, u32 delay, int grid) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int it = 0; it < 2; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_pat<MODE, NT>), dim3(grid), dim3(64), 0, 0, hash, pos, nunits, rows, fill, delay);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)nunits * fill * 64 * 12;
    printf("%-14s grid=%5d fill=%2u/%2u delay=%2u  %.3f ms  %.2f TB/s\n", name, grid, fill, rows, delay, ms, bytes / ms / 1e9);
}

int main() {
    const u32 nunits = 1562500 / 4, rows = 32;  // 2.5e7 reads' worth: 9.6 GB of slabs
    u64 *hash;
    u32 *pos;
    if (hipMalloc(&hash, (size_t)nunits * rows * 64 * 8) != hipSuccess || hipMalloc(&pos, (size_t)nunits * rows * 64 * 4) != hipSuccess) return 1;
    hipMemset(hash, 0, (size_t)nunits * rows * 64 * 8);
    hipDeviceSynchronize();
    for (int grid : {3072, 2048, 8192, 1024}) {
        run<0, true>("rows nt", hash, pos, nunits, rows, 28, 0, grid);
        run<0, false>("rows plain", hash, pos, nunits, rows, 28, 0, grid);
        run<1, true>("dense nt", hash, pos, nunits, rows, 28, 0, grid);
        run<1, false>("dense plain", hash, pos, nunits, rows, 28, 0, grid);
    }
    for (u32 delay : {1, 2, 3, 4}) {
        run<0, true>("rows nt", hash, pos, nunits, rows, 28, delay, 3072);
        run<1, true>("dense nt", hash, pos, nunits, rows, 28, delay, 3072);
    }
    hipDeviceSynchronize();
    return 0;
}
