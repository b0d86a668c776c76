// dev microbenchmark: what does a partial-line write cost in HBM traffic?  Each lane group writes one aligned piece of PIECE bytes
// into its own 128-byte line of a buffer far larger than the caches (no merging possible); FETCH_SIZE / WRITE_SIZE of the
// kernel tell whether the L2 fills the rest of the line first.   hipcc --offload-arch=gfx950 -O2 -o partial_write partial_write.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;

template <int PIECE, bool NT>  // PIECE bytes per line, written by PIECE/8 lanes with 8-byte stores
__global__ void k_piece(u64 *buf, u64 nlines, u32 off8) {
    constexpr int LPL = PIECE / 8;  // lanes per line
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 line = gid / LPL;
    const u32 sub = (u32)(gid % LPL);
    if (line >= nlines) return;
    // scatter the lines so that neighbouring lanes groups do not share DRAM pages trivially
    const u64 l2 = (line * 2654435761ULL) % nlines;
    u64 *p = buf + l2 * 16 + off8 + sub;
    if (NT) __builtin_nontemporal_store(gid, p);
    else *p = gid;
}

template <int PIECE, bool NT>
void run(const char *name, u64 *buf, u64 nlines, u32 off8) {
    constexpr int LPL = PIECE / 8;
    const u64 threads = nlines * LPL;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_piece<PIECE, NT>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, 0, buf, nlines, off8);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s piece=%3d B off=%3u  %.3f ms  %.1f GB/s of payload\n", name, PIECE, off8 * 8, ms, nlines * (double)PIECE / ms / 1e6);
}

int main() {
    const u64 nlines = 64ull << 20;  // 8 GiB buffer
    u64 *buf;
    if (hipMalloc(&buf, nlines * 128) != hipSuccess) return 1;
    hipMemset(buf, 0, nlines * 128);
    hipDeviceSynchronize();
    run<128, false>("plain_128", buf, nlines, 0);
    run<64, false>("plain_64_aligned", buf, nlines, 0);
    run<64, false>("plain_64_off32", buf, nlines, 4);
    run<32, false>("plain_32_aligned", buf, nlines, 0);
    run<32, false>("plain_32_off16", buf, nlines, 2);
    run<16, false>("plain_16", buf, nlines, 0);
    run<8, false>("plain_8", buf, nlines, 0);
    run<128, true>("nt_128", buf, nlines, 0);
    run<64, true>("nt_64_aligned", buf, nlines, 0);
    run<32, true>("nt_32_aligned", buf, nlines, 0);
    run<16, true>("nt_16", buf, nlines, 0);
    hipDeviceSynchronize();
    return 0;
}
