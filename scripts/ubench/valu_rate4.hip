// dev microbenchmark #4: SDWA byte selects, bit-field ops, 64-bit compares (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32;
typedef unsigned long long u64;
#define REP8(X) X X X X X X X X
#define R32 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)
#define R64 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(c)
template <int OP>
__global__ __launch_bounds__(64) void k(u32 *out, int iters, u32 seed) {
    u64 x0 = threadIdx.x + seed, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7;
    u32 a0 = (u32)x0, a1 = (u32)x1, a2 = (u32)x2, a3 = (u32)x3, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    u32 b = (seed * 77 + 1) | 0x10001, c = seed + 5;
    u64 msk = 0x5555aaaa0f0f3333ull * seed;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) {
            REP8(asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n"
 "v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n" R32);)
        }
        if (OP == 1) {
            REP8(asm volatile("v_and_b32_sdwa %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_and_b32_sdwa %1, %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_and_b32_sdwa %2, %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_and_b32_sdwa %3, %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
 "v_and_b32_sdwa %4, %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_and_b32_sdwa %5, %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_and_b32_sdwa %6, %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_and_b32_sdwa %7, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n" R32);)
        }
        if (OP == 2) {
            REP8(asm volatile("v_mov_b32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n v_mov_b32_sdwa %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n v_mov_b32_sdwa %2, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n v_mov_b32_sdwa %3, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n"
 "v_mov_b32_sdwa %4, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n v_mov_b32_sdwa %5, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n v_mov_b32_sdwa %6, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n v_mov_b32_sdwa %7, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n" R32);)
        }
        if (OP == 3) {
            REP8(asm volatile("v_bfe_u32 %0, %0, 4, 8\n v_bfe_u32 %1, %1, 4, 8\n v_bfe_u32 %2, %2, 4, 8\n v_bfe_u32 %3, %3, 4, 8\n"
 "v_bfe_u32 %4, %4, 4, 8\n v_bfe_u32 %5, %5, 4, 8\n v_bfe_u32 %6, %6, 4, 8\n v_bfe_u32 %7, %7, 4, 8\n" R32);)
        }
        if (OP == 4) {
            REP8(asm volatile("v_bfi_b32 %0, %8, %0, %9\n v_bfi_b32 %1, %8, %1, %9\n v_bfi_b32 %2, %8, %2, %9\n v_bfi_b32 %3, %8, %3, %9\n"
 "v_bfi_b32 %4, %8, %4, %9\n v_bfi_b32 %5, %8, %5, %9\n v_bfi_b32 %6, %8, %6, %9\n v_bfi_b32 %7, %8, %7, %9\n" R32);)
        }
        if (OP == 5) {
            REP8(asm volatile("v_and_or_b32 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %8, %9\n v_and_or_b32 %2, %2, %8, %9\n v_and_or_b32 %3, %3, %8, %9\n"
 "v_and_or_b32 %4, %4, %8, %9\n v_and_or_b32 %5, %5, %8, %9\n v_and_or_b32 %6, %6, %8, %9\n v_and_or_b32 %7, %7, %8, %9\n" R32);)
        }
        if (OP == 6) {
            REP8(asm volatile("v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n"
 "v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9\n" R32);)
        }
        if (OP == 7) {
            REP8(asm volatile("v_alignbit_b32 %0, %0, %8, 31\n v_alignbit_b32 %1, %1, %8, 31\n v_alignbit_b32 %2, %2, %8, 31\n v_alignbit_b32 %3, %3, %8, 31\n"
 "v_alignbit_b32 %4, %4, %8, 31\n v_alignbit_b32 %5, %5, %8, 31\n v_alignbit_b32 %6, %6, %8, 31\n v_alignbit_b32 %7, %7, %8, 31\n" R32);)
        }
        if (OP == 8) {
            REP8(asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
 "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n" R32);)
        }
        if (OP == 9) {
            REP8(asm volatile("v_cmp_lt_u64 s[20:21], %0, %1\n v_cmp_lt_u64 s[22:23], %1, %2\n v_cmp_lt_u64 s[24:25], %2, %3\n v_cmp_lt_u64 s[26:27], %3, %0\n"
                              "v_cmp_lt_u64 s[20:21], %0, %2\n v_cmp_lt_u64 s[22:23], %1, %3\n v_cmp_lt_u64 s[24:25], %2, %0\n v_cmp_lt_u64 s[26:27], %3, %1\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : : "s20","s21","s22","s23","s24","s25","s26","s27");)
        }
        if (OP == 10) {
            REP8(asm volatile("v_cndmask_b32 %0, %0, %8, %10\n v_cndmask_b32 %1, %1, %8, %10\n v_cndmask_b32 %2, %2, %8, %10\n v_cndmask_b32 %3, %3, %8, %10\n"
                              "v_cndmask_b32 %4, %4, %9, %10\n v_cndmask_b32 %5, %5, %9, %10\n v_cndmask_b32 %6, %6, %9, %10\n v_cndmask_b32 %7, %7, %9, %10\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(msk));)
        }
        if (OP == 11) {
            REP8(asm volatile("v_bitop3_b32 %0, %0, %8, %9 bitop3:0x96\n v_bitop3_b32 %1, %1, %8, %9 bitop3:0x96\n v_bitop3_b32 %2, %2, %8, %9 bitop3:0x96\n v_bitop3_b32 %3, %3, %8, %9 bitop3:0x96\n"
 "v_bitop3_b32 %4, %4, %8, %9 bitop3:0x96\n v_bitop3_b32 %5, %5, %8, %9 bitop3:0x96\n v_bitop3_b32 %6, %6, %8, %9 bitop3:0x96\n v_bitop3_b32 %7, %7, %8, %9 bitop3:0x96\n" R32);)
        }
        if (OP == 12) {
            REP8(asm volatile("v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 3, %3\n"
 "v_lshrrev_b32 %4, 3, %4\n v_lshrrev_b32 %5, 3, %5\n v_lshrrev_b32 %6, 3, %6\n v_lshrrev_b32 %7, 3, %7\n" R32);)
        }
        if (OP == 13) {
            REP8(asm volatile("v_lshl_or_b32 %0, %0, 4, %8\n v_lshl_or_b32 %1, %1, 4, %8\n v_lshl_or_b32 %2, %2, 4, %8\n v_lshl_or_b32 %3, %3, 4, %8\n"
 "v_lshl_or_b32 %4, %4, 4, %8\n v_lshl_or_b32 %5, %5, 4, %8\n v_lshl_or_b32 %6, %6, 4, %8\n v_lshl_or_b32 %7, %7, 4, %8\n" R32);)
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = (u32)(x0 ^ x1 ^ x2 ^ x3) ^ (u32)((x0 ^ x1 ^ x2 ^ x3) >> 32) ^ a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
template <int OP>
void run(const char *name, u32 *d, int wpc, int per_iter = 64) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256 * wpc;
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(64), 0, 0, d, 10, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(64), 0, 0, d, iters, 1u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)grid * iters * per_iter;
    // cycles per wave-instruction per SIMD at 1.92 GHz (the clock measured under integer load)
    printf("%-22s waves/CU=%2d  %.3f ms => %.2f cycles/inst/SIMD @1.92GHz\n", name, wpc, ms, (ms * 1e-3 * 1.92e9) / (insts / 1024.0));
}
int main() {
    u32 *d; hipMalloc(&d, 256 * 32 * 64 * 4);
    for (int wpc : {4, 8, 16}) {
        run<0>("v_and_b32_e32", d, wpc);
        run<1>("v_and_b32_sdwa BYTE_1", d, wpc);
        run<2>("v_mov_b32_sdwa BYTE_2", d, wpc);
        run<3>("v_bfe_u32", d, wpc);
        run<4>("v_bfi_b32", d, wpc);
        run<5>("v_and_or_b32", d, wpc);
        run<6>("v_perm_b32", d, wpc);
        run<7>("v_alignbit_b32", d, wpc);
        run<8>("v_xor_b32_e32", d, wpc);
        run<9>("v_cmp_lt_u64 (sgpr)", d, wpc);
        run<10>("v_cndmask_e64 sgpr", d, wpc);
        run<11>("v_bitop3_b32", d, wpc);
        run<12>("v_lshrrev_b32_e32", d, wpc);
        run<13>("v_lshl_or_b32", d, wpc);
    }
    return 0;
}
