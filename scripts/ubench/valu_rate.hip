// dev microbenchmark: issue rate of the integer VALU ops the sketch kernels are made of (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32;
typedef unsigned long long u64;

#define REP8(X) X X X X X X X X
template <int OP>
__global__ __launch_bounds__(64) void k(u32 *out, int iters, u32 seed) {
    u32 a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    u32 b = seed * 77 + 1, c = seed + 5;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) {  // v_xor_b32
            REP8(asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
                              "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if (OP == 1) {  // v_alignbit_b32
            REP8(asm volatile("v_alignbit_b32 %0, %0, %1, 31\n v_alignbit_b32 %1, %1, %2, 31\n v_alignbit_b32 %2, %2, %3, 31\n v_alignbit_b32 %3, %3, %4, 31\n"
                              "v_alignbit_b32 %4, %4, %5, 31\n v_alignbit_b32 %5, %5, %6, 31\n v_alignbit_b32 %6, %6, %7, 31\n v_alignbit_b32 %7, %7, %0, 31\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (OP == 2) {  // v_cndmask_b32 (vcc)
            asm volatile("v_cmp_lt_u32 vcc, %0, %1" ::"v"(a0), "v"(b) : "vcc");
            REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                              "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
        } else if (OP == 3) {  // v_cmp_lt_u64 (to vcc)
            REP8(asm volatile("v_cmp_lt_u64 vcc, %0, %1\n v_cmp_lt_u64 vcc, %1, %2\n v_cmp_lt_u64 vcc, %2, %3\n v_cmp_lt_u64 vcc, %3, %0\n"
                              "v_cmp_lt_u64 vcc, %0, %2\n v_cmp_lt_u64 vcc, %1, %3\n v_cmp_lt_u64 vcc, %2, %0\n v_cmp_lt_u64 vcc, %3, %1\n"
                              ::"v"(((u64)a0 << 32) | a1), "v"(((u64)a2 << 32) | a3), "v"(((u64)a4 << 32) | a5), "v"(((u64)a6 << 32) | a7) : "vcc");)
        } else if (OP == 4) {  // v_cmp_lt_u32
            REP8(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %2\n v_cmp_lt_u32 vcc, %2, %3\n v_cmp_lt_u32 vcc, %3, %0\n"
                              "v_cmp_lt_u32 vcc, %0, %2\n v_cmp_lt_u32 vcc, %1, %3\n v_cmp_lt_u32 vcc, %2, %0\n v_cmp_lt_u32 vcc, %3, %1\n"
                              ::"v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");)
        } else if (OP == 5) {  // v_and_or_b32
            REP8(asm volatile("v_and_or_b32 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %8, %9\n v_and_or_b32 %2, %2, %8, %9\n v_and_or_b32 %3, %3, %8, %9\n"
                              "v_and_or_b32 %4, %4, %8, %9\n v_and_or_b32 %5, %5, %8, %9\n v_and_or_b32 %6, %6, %8, %9\n v_and_or_b32 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (OP == 6) {  // v_add_u32
            REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                              "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if (OP == 7) {  // v_fma_f32
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (OP == 8) {  // v_cndmask_b32 e64 with SGPR-pair mask
            u64 m = __ballot(a0 < b);
            REP8(asm volatile("v_cndmask_b32 %0, %0, %8, %9\n v_cndmask_b32 %1, %1, %8, %9\n v_cndmask_b32 %2, %2, %8, %9\n v_cndmask_b32 %3, %3, %8, %9\n"
                              "v_cndmask_b32 %4, %4, %8, %9\n v_cndmask_b32 %5, %5, %8, %9\n v_cndmask_b32 %6, %6, %8, %9\n v_cndmask_b32 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(m));)
        } else if (OP == 9) {  // dependent chain v_xor (latency)
            REP8(asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n"
                              "v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n"
                              : "+v"(a0) : "v"(b));)
        } else if (OP == 10) {  // v_lshl_add_u32
            REP8(asm volatile("v_lshl_add_u32 %0, %0, 1, %8\n v_lshl_add_u32 %1, %1, 1, %8\n v_lshl_add_u32 %2, %2, 1, %8\n v_lshl_add_u32 %3, %3, 1, %8\n"
                              "v_lshl_add_u32 %4, %4, 1, %8\n v_lshl_add_u32 %5, %5, 1, %8\n v_lshl_add_u32 %6, %6, 1, %8\n v_lshl_add_u32 %7, %7, 1, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if (OP == 11) {  // v_mad_u64_u32
            u64 x = ((u64)a0 << 32) | a1, y = ((u64)a2 << 32) | a3;
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n"
                              "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n"
                              : "+v"(x), "+v"(y) : "v"(b), "v"(c) : "vcc");)
            a0 ^= (u32)x ^ (u32)y;
        } else if (OP == 12) {  // v_mul_lo_u32
            REP8(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                              "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if (OP == 13) {  // v_pk_add_u16 (packed)
            REP8(asm volatile("v_pk_add_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %8\n v_pk_add_u16 %2, %2, %8\n v_pk_add_u16 %3, %3, %8\n"
                              "v_pk_add_u16 %4, %4, %8\n v_pk_add_u16 %5, %5, %8\n v_pk_add_u16 %6, %6, %8\n v_pk_add_u16 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if (OP == 14) {  // v_bfi_b32
            REP8(asm volatile("v_bfi_b32 %0, %8, %0, %9\n v_bfi_b32 %1, %8, %1, %9\n v_bfi_b32 %2, %8, %2, %9\n v_bfi_b32 %3, %8, %3, %9\n"
                              "v_bfi_b32 %4, %8, %4, %9\n v_bfi_b32 %5, %8, %5, %9\n v_bfi_b32 %6, %8, %6, %9\n v_bfi_b32 %7, %8, %7, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <int OP>
void run(const char *name, u32 *d, int wpc) {
    const int iters = 2000, per_iter = 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    int grid = 256 * wpc;
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(64), 0, 0, d, 10, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(64), 0, 0, d, iters, 1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)grid * iters * per_iter;  // wave-instructions
    double cyc_per_inst_per_simd = (ms * 1e-3 * 2.4e9) / (insts / (256.0 * 4));
    printf("%-22s waves/CU=%2d  %.3f ms  %.1f G wave-inst/s  => %.2f cycles/inst/SIMD @2.4GHz\n", name, wpc, ms, insts / ms / 1e6,
           cyc_per_inst_per_simd);
}

int main() {
    u32 *d;
    hipMalloc(&d, 256 * 32 * 64 * 4);
    for (int wpc : {4, 8, 16, 32}) {
        run<0>("v_xor_b32", d, wpc);
        run<1>("v_alignbit_b32", d, wpc);
        run<2>("v_cndmask_b32 vcc", d, wpc);
        run<8>("v_cndmask_b32 sgpr", d, wpc);
        run<3>("v_cmp_lt_u64", d, wpc);
        run<4>("v_cmp_lt_u32", d, wpc);
        run<5>("v_and_or_b32", d, wpc);
        run<6>("v_add_u32", d, wpc);
        run<10>("v_lshl_add_u32", d, wpc);
        run<14>("v_bfi_b32", d, wpc);
        run<7>("v_fma_f32", d, wpc);
        run<13>("v_pk_add_u16", d, wpc);
        run<12>("v_mul_lo_u32", d, wpc);
        run<11>("v_mad_u64_u32", d, wpc);
        run<9>("v_xor dependent", d, wpc);
        printf("\n");
    }
    return 0;
}
