// Ceiling of the sweep's memory pattern: coalesced tile reads, writes as 256 runs of RUN elements per tile.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <typename T, int KPT>
__global__ __launch_bounds__(512) void scat(const T* __restrict__ src, T* __restrict__ dst, size_t n, int run_log, int xcd_major, int skew) {
    constexpr int TILE = 512 * KPT;
    const size_t ntiles = n / TILE;
    const size_t per_bucket = n / 256;
    // xcd_major: consecutive tiles are handled by workgroups of ONE XCD (blocks are dealt round-robin over 8 XCDs)
    const size_t first = xcd_major ? (blockIdx.x % 8) * (ntiles / 8) + blockIdx.x / 8 : blockIdx.x;
    const size_t step = xcd_major ? gridDim.x / 8 : gridDim.x;
    const size_t last = xcd_major ? (blockIdx.x % 8 + 1) * (ntiles / 8) : ntiles;
    for (size_t t = first; t < last; t += step) {
        T v[KPT];
        const T* p = src + t * TILE + threadIdx.x;
#pragma unroll
        for (int i = 0; i < KPT; ++i) v[i] = p[i * 512];
        const int run = 1 << run_log;  // elements per digit per tile = TILE/256
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const unsigned slot = i * 512 + threadIdx.x;
            const unsigned d = slot >> run_log;
            dst[d * per_bucket + t * run + (slot & (run - 1)) + skew] = v[i];
        }
    }
}
template <typename T, int KPT> void run(const char* name, size_t bytes, int occ, int xm, int skew) {
    const size_t n = bytes / sizeof(T);
    T *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes + 4096));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int run_log = 0; while ((1 << run_log) < 512 * KPT / 256) ++run_log;
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((scat<T, KPT>), dim3(256 * occ), dim3(512), 0, 0, a, b, n, run_log, xm, skew);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((scat<T, KPT>), dim3(256 * occ), dim3(512), 0, 0, a, b, n, run_log, xm, skew);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    printf("%-34s xcd-major %d skew %2d occ %d: %.3f ms  %.0f GB/s (read+write)\n", name, xm, skew, occ, ms, 2.0 * bytes / ms / 1e6);
    CK(hipFree(a)); CK(hipFree(b));
}
int main() {
    const size_t bytes = 1ull << 30;
    for (int xm : {0, 1}) for (int skew : {0, 13}) for (int occ : {2, 4}) {
        run<unsigned, 16>("u32 x16 (runs of 128 B)", bytes, occ, xm, skew);
        run<unsigned long long, 8>("u64 x8 (runs of 128 B)", bytes, occ, xm, skew);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
