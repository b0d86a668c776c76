// dev microbenchmark #5: the 64-bit "conditional move of three words" idiom (sliding-minimum update), three encodings (gfx950).
//   X: v_cmp_lt_u64 -> SGPR pair, 3 x v_cndmask_b32_e64
//   Y: v_cmp_lt_u64 -> SGPR pair, s_mov exec, 3 x v_mov_b32 (masked), s_mov exec,-1
//   Z: v_cmpx_lt_u64 (writes EXEC), 3 x v_mov_b32 (masked), s_mov exec,-1
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32;
typedef unsigned long long u64;
#define REP8(X) X X X X X X X X
template <int OP>
__global__ __launch_bounds__(64) void k(u32 *out, int iters, u32 seed) {
    u64 x0 = threadIdx.x * 0x9E3779B97F4A7C15ull + seed, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7;
    u32 p0 = threadIdx.x, p1 = p0 * 3, p2 = p0 * 5, p3 = p0 * 7;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) {
            REP8({ bool c = x1 < x0; x0 = c ? x1 : x0; p0 = c ? p1 : p0; c = x3 < x2; x2 = c ? x3 : x2; p2 = c ? p3 : p2;
                   c = x2 < x1; x1 = c ? x2 : x1; p1 = c ? p2 : p1; c = x0 < x3; x3 = c ? x0 : x3; p3 = c ? p0 : p3;
                   asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
                   x0 += 0x1234567; x2 ^= x1 >> 3; })
        } else if (OP == 1) {
            REP8(asm volatile("v_cmp_lt_u64 s[20:21], %1, %0\n s_mov_b64 exec, s[20:21]\n v_mov_b64 %0, %1\n v_mov_b32 %4, %5\n s_mov_b64 exec, -1\n"
                              "v_cmp_lt_u64 s[22:23], %3, %2\n s_mov_b64 exec, s[22:23]\n v_mov_b64 %2, %3\n v_mov_b32 %6, %7\n s_mov_b64 exec, -1\n"
                              "v_cmp_lt_u64 s[20:21], %2, %1\n s_mov_b64 exec, s[20:21]\n v_mov_b64 %1, %2\n v_mov_b32 %5, %6\n s_mov_b64 exec, -1\n"
                              "v_cmp_lt_u64 s[22:23], %0, %3\n s_mov_b64 exec, s[22:23]\n v_mov_b64 %3, %0\n v_mov_b32 %7, %4\n s_mov_b64 exec, -1\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : : "s20", "s21", "s22", "s23");
                 x0 += 0x1234567; x2 ^= x1 >> 3;)
        } else if (OP == 4) {  // the kernels' form: ballot compare -> SGPR pair, three VOP3 selects
#define SEL(m, t, f) ({ u32 r_; asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r_) : "v"(f), "v"(t), "s"(m)); r_; })
#define MINOP(a, b, pa, pb) { u64 m_ = __builtin_amdgcn_ballot_w64(b < a); u32 lo_ = SEL(m_, (u32)b, (u32)a), hi_ = SEL(m_, (u32)(b >> 32), (u32)(a >> 32)); pa = SEL(m_, pb, pa); a = ((u64)hi_ << 32) | lo_; }
            REP8({ MINOP(x0, x1, p0, p1) MINOP(x2, x3, p2, p3) MINOP(x1, x2, p1, p2) MINOP(x3, x0, p3, p0)
                   asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
                   x0 += 0x1234567; x2 ^= x1 >> 3; })
        } else if (OP == 5) {  // only the filler ops (subtract from the others)
            REP8({ asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
                   x0 += 0x1234567; x2 ^= x1 >> 3; })
        } else if (OP == 3) {  // same with v_mov_b64 for the hash pair
            REP8(asm volatile("v_cmpx_lt_u64 vcc, %1, %0\n v_mov_b64 %0, %1\n v_mov_b32 %4, %5\n s_mov_b64 exec, -1\n"
                              "v_cmpx_lt_u64 vcc, %3, %2\n v_mov_b64 %2, %3\n v_mov_b32 %6, %7\n s_mov_b64 exec, -1\n"
                              "v_cmpx_lt_u64 vcc, %2, %1\n v_mov_b64 %1, %2\n v_mov_b32 %5, %6\n s_mov_b64 exec, -1\n"
                              "v_cmpx_lt_u64 vcc, %0, %3\n v_mov_b64 %3, %0\n v_mov_b32 %7, %4\n s_mov_b64 exec, -1\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : : "vcc");
                 x0 += 0x1234567; x2 ^= x1 >> 3;)
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = (u32)(x0 ^ x1 ^ x2 ^ x3) ^ (u32)((x0 ^ x1 ^ x2 ^ x3) >> 32) ^ p0 ^ p1 ^ p2 ^ p3;
}
template <int OP>
void run(const char *name, u32 *d, u32 *h, int wpc) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256 * wpc;
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(64), 0, 0, d, 10, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(64), 0, 0, d, iters, 1u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, d, 64 * 4, hipMemcpyDeviceToHost);
    u32 cs = 0; for (int i = 0; i < 64; ++i) cs = cs * 31 + h[i];
    double ops = (double)grid * iters * 32;  // min-ops
    printf("%-34s waves/CU=%2d  %.3f ms => %.2f cycles per 3-word conditional move per SIMD @1.92GHz  (check %08x)\n", name, wpc, ms, (ms * 1e-3 * 1.92e9) / (ops / 1024.0), cs);
}
int main() {
    u32 *d; hipMalloc(&d, 256 * 32 * 64 * 4);
    u32 h[64];
    for (int wpc : {4, 8, 16}) {
        run<0>("X compiler: cmp + 3 cndmask_e32 vcc", d, h, wpc); run<4>("X64 cmp_e64 + 3 cndmask_e64", d, h, wpc); run<5>("(filler only)", d, h, wpc); run<1>("Y cmp + s_mov exec + mov_b64 + mov", d, h, wpc);
        run<3>("Z' cmpx + mov_b64 + mov", d, h, wpc);
    }
    return 0;
}
