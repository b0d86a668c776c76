// dev microbenchmark #2: more VALU opcodes (gfx950), 8 waves/CU and 16 waves/CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32;
typedef unsigned long long u64;
#define REP8(X) X X X X X X X X
#define OP2(name) \
    name " %0, %0, %8\n " name " %1, %1, %8\n " name " %2, %2, %8\n " name " %3, %3, %8\n " name " %4, %4, %8\n " name " %5, %5, %8\n " name " %6, %6, %8\n " name " %7, %7, %8\n"
#define OP3(name) \
    name " %0, %0, %8, %9\n " name " %1, %1, %8, %9\n " name " %2, %2, %8, %9\n " name " %3, %3, %8, %9\n " name " %4, %4, %8, %9\n " name " %5, %5, %8, %9\n " name " %6, %6, %8, %9\n " name " %7, %7, %8, %9\n"
#define OPREV(name) \
    name " %0, %8, %0\n " name " %1, %8, %1\n " name " %2, %8, %2\n " name " %3, %8, %3\n " name " %4, %8, %4\n " name " %5, %8, %5\n " name " %6, %8, %6\n " name " %7, %8, %7\n"
#define REGS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)

template <int OP>
__global__ __launch_bounds__(64) void k(u32 *out, int iters, u32 seed) {
    u32 a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    u32 b = (seed * 77 + 1) & 7, c = seed + 5;
    asm volatile("v_cmp_lt_u32 vcc, %0, %1" ::"v"(a0), "v"(c) : "vcc");
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP8(asm volatile(OP2("v_and_b32") REGS);) }
        else if (OP == 1) { REP8(asm volatile(OP2("v_or_b32") REGS);) }
        else if (OP == 2) { REP8(asm volatile(OPREV("v_lshlrev_b32") REGS);) }
        else if (OP == 3) { REP8(asm volatile(OPREV("v_lshrrev_b32") REGS);) }
        else if (OP == 4) { REP8(asm volatile(OP2("v_min_u32") REGS);) }
        else if (OP == 5) { REP8(asm volatile(OP2("v_sub_u32") REGS);) }
        else if (OP == 6) { REP8(asm volatile(OP3("v_bfe_u32") REGS);) }
        else if (OP == 7) { REP8(asm volatile(OP3("v_lshl_or_b32") REGS);) }
        else if (OP == 8) { REP8(asm volatile(OP3("v_or3_b32") REGS);) }
        else if (OP == 9) { REP8(asm volatile(OP3("v_add3_u32") REGS);) }
        else if (OP == 10) { REP8(asm volatile(OP3("v_xad_u32") REGS);) }
        else if (OP == 11) { REP8(asm volatile(OP3("v_mad_u32_u24") REGS);) }
        else if (OP == 12) { REP8(asm volatile(OP3("v_perm_b32") REGS);) }
        else if (OP == 13) {  // v_cndmask e32, vcc set once outside the loop
            REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                              "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n" REGS);)
        } else if (OP == 14) {  // v_addc_co_u32 (vcc in/out)
            REP8(asm volatile("v_addc_co_u32 %0, vcc, %0, %8, vcc\n v_addc_co_u32 %1, vcc, %1, %8, vcc\n v_addc_co_u32 %2, vcc, %2, %8, vcc\n v_addc_co_u32 %3, vcc, %3, %8, vcc\n"
                              "v_addc_co_u32 %4, vcc, %4, %8, vcc\n v_addc_co_u32 %5, vcc, %5, %8, vcc\n v_addc_co_u32 %6, vcc, %6, %8, vcc\n v_addc_co_u32 %7, vcc, %7, %8, vcc\n" REGS : "vcc");)
        } else if (OP == 15) {  // v_mov_b32
            REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n" REGS);)
        } else if (OP == 16) {  // v_cmp_lt_u64 e64 to sgpr pair + v_cndmask e64 pair (the min64 idiom: 3 insts)
            u64 x = ((u64)a0 << 32) | a1, y = ((u64)a2 << 32) | a3;
            REP8(asm volatile("v_cmp_lt_u64 s[20:21], %0, %1\n v_cndmask_b32 %2, %2, %3, s[20:21]\n v_cndmask_b32 %3, %3, %2, s[20:21]\n v_cmp_lt_u64 s[22:23], %1, %0\n v_cndmask_b32 %4, %4, %5, s[22:23]\n v_cndmask_b32 %5, %5, %4, s[22:23]\n"
                              "v_cmp_lt_u64 s[24:25], %0, %1\n v_cndmask_b32 %2, %2, %3, s[24:25]\n"
                              : "+v"(x), "+v"(y), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "s20", "s21", "s22", "s23", "s24", "s25");)
            a0 ^= (u32)x; a2 ^= (u32)y;
        } else if (OP == 17) { REP8(asm volatile(OP2("v_max_u32") REGS);) }
        else if (OP == 18) { REP8(asm volatile(OP3("v_alignbit_b32") REGS);) }
        else if (OP == 19) {  // v_mbcnt_lo_u32_b32
            REP8(asm volatile(OP2("v_mbcnt_lo_u32_b32") REGS);) }
        else if (OP == 20) {  // ds_read_b128 throughput (random-ish 16-entry table)
            __shared__ uint4 tab[64];
            tab[threadIdx.x] = make_uint4(a0, a1, a2, a3);
            __syncthreads();
            for (int r = 0; r < 16; ++r) { uint4 v = tab[(a0 >> r) & 15]; a1 ^= v.x; a2 ^= v.y; a3 ^= v.z; a4 ^= v.w; }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
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
    printf("%-26s waves/CU=%2d  %.3f ms => %.2f cycles/inst/SIMD @2.4GHz\n", name, wpc, ms, (ms * 1e-3 * 2.4e9) / (insts / 1024.0));
}
int main() {
    u32 *d; hipMalloc(&d, 256 * 32 * 64 * 4);
    for (int wpc : {8, 16}) {
        run<0>("v_and_b32", d, wpc); run<1>("v_or_b32", d, wpc); run<2>("v_lshlrev_b32", d, wpc); run<3>("v_lshrrev_b32", d, wpc);
        run<4>("v_min_u32", d, wpc); run<17>("v_max_u32", d, wpc); run<5>("v_sub_u32", d, wpc); run<15>("v_mov_b32", d, wpc);
        run<6>("v_bfe_u32", d, wpc); run<7>("v_lshl_or_b32", d, wpc); run<8>("v_or3_b32", d, wpc); run<9>("v_add3_u32", d, wpc);
        run<10>("v_xad_u32", d, wpc); run<11>("v_mad_u32_u24", d, wpc); run<12>("v_perm_b32", d, wpc); run<18>("v_alignbit_b32 (vgpr sh)", d, wpc);
        run<13>("v_cndmask_b32 e32 vcc", d, wpc); run<14>("v_addc_co_u32", d, wpc); run<19>("v_mbcnt_lo", d, wpc);
        run<16>("min64 idiom (cmp+2cnd)", d, wpc); run<20>("ds_read_b128+4xor (x16)", d, wpc, 16);
        printf("\n");
    }
    return 0;
}
