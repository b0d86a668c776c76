// dev microbenchmark #3: 64-bit / multiply VALU opcodes (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32;
typedef unsigned long long u64;
#define REP8(X) X X X X X X X X
#define R32 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)
#define R64 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(c)
template <int OP>
__global__ __launch_bounds__(64) void k(u32 *out, int iters, u32 seed) {
    u64 x0 = threadIdx.x + seed, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7, x4 = x0 * 11, x5 = x0 * 13, x6 = x0 * 17, x7 = x0 * 19;
    u32 a0 = (u32)x0, a1 = (u32)x1, a2 = (u32)x2, a3 = (u32)x3, a4 = (u32)x4, a5 = (u32)x5, a6 = (u32)x6, a7 = (u32)x7;
    u32 b = (seed * 77 + 1) | 0x10001, c = seed + 5;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) {  // v_mad_u64_u32 d64 = lo(d)*b + d64
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                              "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n" R64 : "vcc");)
        } else if (OP == 1) {  // v_lshl_add_u64
            REP8(asm volatile("v_lshl_add_u64 %0, %0, 1, %1\n v_lshl_add_u64 %1, %1, 1, %2\n v_lshl_add_u64 %2, %2, 1, %3\n v_lshl_add_u64 %3, %3, 1, %4\n"
                              "v_lshl_add_u64 %4, %4, 1, %5\n v_lshl_add_u64 %5, %5, 1, %6\n v_lshl_add_u64 %6, %6, 1, %7\n v_lshl_add_u64 %7, %7, 1, %0\n" R64);)
        } else if (OP == 2) {  // v_mul_lo_u32
            REP8(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                              "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n" R32);)
        } else if (OP == 3) {  // v_mul_hi_u32
            REP8(asm volatile("v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n"
                              "v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8\n" R32);)
        } else if (OP == 4) {  // v_mul_u32_u24
            REP8(asm volatile("v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n"
                              "v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8\n" R32);)
        } else if (OP == 5) {  // add_co + addc_co pair (64-bit add)
            REP8(asm volatile("v_add_co_u32 %0, vcc, %0, %8\n v_addc_co_u32 %1, vcc, %1, %9, vcc\n v_add_co_u32 %2, vcc, %2, %8\n v_addc_co_u32 %3, vcc, %3, %9, vcc\n"
                              "v_add_co_u32 %4, vcc, %4, %8\n v_addc_co_u32 %5, vcc, %5, %9, vcc\n v_add_co_u32 %6, vcc, %6, %8\n v_addc_co_u32 %7, vcc, %7, %9, vcc\n" R32 : "vcc");)
        } else if (OP == 6) {  // v_lshlrev_b64
            REP8(asm volatile("v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 1, %1\n v_lshlrev_b64 %2, 1, %2\n v_lshlrev_b64 %3, 1, %3\n"
                              "v_lshlrev_b64 %4, 1, %4\n v_lshlrev_b64 %5, 1, %5\n v_lshlrev_b64 %6, 1, %6\n v_lshlrev_b64 %7, 1, %7\n" R64);)
        } else if (OP == 7) {  // v_mul_hi_u32_u24
            REP8(asm volatile("v_mul_hi_u32_u24 %0, %0, %8\n v_mul_hi_u32_u24 %1, %1, %8\n v_mul_hi_u32_u24 %2, %2, %8\n v_mul_hi_u32_u24 %3, %3, %8\n"
                              "v_mul_hi_u32_u24 %4, %4, %8\n v_mul_hi_u32_u24 %5, %5, %8\n v_mul_hi_u32_u24 %6, %6, %8\n v_mul_hi_u32_u24 %7, %7, %8\n" R32);)
        } else if (OP == 8) {  // v_mad_u32_u16 ? -> use v_mad_u32_u24 as reference
            REP8(asm volatile("v_mad_u32_u24 %0, %0, %8, %9\n v_mad_u32_u24 %1, %1, %8, %9\n v_mad_u32_u24 %2, %2, %8, %9\n v_mad_u32_u24 %3, %3, %8, %9\n"
                              "v_mad_u32_u24 %4, %4, %8, %9\n v_mad_u32_u24 %5, %5, %8, %9\n v_mad_u32_u24 %6, %6, %8, %9\n v_mad_u32_u24 %7, %7, %8, %9\n" R32);)
        } else if (OP == 9) {  // v_pk_mul_lo_u16 (2 x 16-bit multiplies per lane)
            REP8(asm volatile("v_pk_mul_lo_u16 %0, %0, %8\n v_pk_mul_lo_u16 %1, %1, %8\n v_pk_mul_lo_u16 %2, %2, %8\n v_pk_mul_lo_u16 %3, %3, %8\n"
                              "v_pk_mul_lo_u16 %4, %4, %8\n v_pk_mul_lo_u16 %5, %5, %8\n v_pk_mul_lo_u16 %6, %6, %8\n v_pk_mul_lo_u16 %7, %7, %8\n" R32);)
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = (u32)(x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7) ^ (u32)((x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7) >> 32) ^ a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
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
    for (int wpc : {8, 16}) {
        run<0>("v_mad_u64_u32", d, wpc); run<1>("v_lshl_add_u64", d, wpc); run<2>("v_mul_lo_u32", d, wpc); run<3>("v_mul_hi_u32", d, wpc);
        run<4>("v_mul_u32_u24", d, wpc); run<5>("add_co+addc_co", d, wpc); run<6>("v_lshlrev_b64", d, wpc); run<7>("v_mul_hi_u32_u24", d, wpc);
        run<8>("v_mad_u32_u24", d, wpc); run<9>("v_pk_mul_lo_u16", d, wpc);
    }
    return 0;
}
