// dev microbenchmark: pure-write and copy bandwidth (calibrates what "HBM bound" means for write-dominated kernels)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
__global__ void k_fill(ulonglong2 *p, size_t n, u64 v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_ulonglong2(v + i, v);
}
__global__ void k_copy(const ulonglong2 *s, ulonglong2 *d, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
__global__ void k_read(const ulonglong2 *s, size_t n, u64 *out) {
    u64 a = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { ulonglong2 v = s[i]; a += v.x ^ v.y; }
    if (a == 0x1234567) out[0] = a;
}
int main() {
    const size_t bytes = 12ull << 30, n = bytes / 16;
    ulonglong2 *a, *b; u64 *o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {2048, 8192, 32768}) {
        float ms;
        k_fill<<<grid, 256>>>(a, n, 1); hipEventRecord(e0); k_fill<<<grid, 256>>>(a, n, 2); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("fill  grid=%5d  %.3f ms  %.2f TB/s (write)\n", grid, ms, bytes / ms / 1e9);
        k_copy<<<grid, 256>>>(a, b, n); hipEventRecord(e0); k_copy<<<grid, 256>>>(a, b, n); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("copy  grid=%5d  %.3f ms  %.2f TB/s (read+write)\n", grid, ms, 2.0 * bytes / ms / 1e9);
        k_read<<<grid, 256>>>(a, n, o); hipEventRecord(e0); k_read<<<grid, 256>>>(a, n, o); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("read  grid=%5d  %.3f ms  %.2f TB/s (read)\n", grid, ms, bytes / ms / 1e9);
    }
    return 0;
}
