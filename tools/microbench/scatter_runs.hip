// How does the sweep's write pattern (256 runs per tile, each RUN elements, unaligned) scale with the run length?
// Coalesced tile reads; thread t writes slot i*WG+t of the tile; slot -> (digit, offset in run).  No compute.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <typename T, int KPT, int WG>
__global__ __launch_bounds__(WG) void scat(const T* __restrict__ src, T* __restrict__ dst, size_t n, int skew, int seq) {
    constexpr int TILE = WG * KPT;
    constexpr int RUN = TILE / 256;
    const size_t ntiles = n / TILE;
    const size_t per_bucket = ntiles * RUN;
    const size_t first = (blockIdx.x % 8) * (ntiles / 8) + blockIdx.x / 8;
    const size_t step = gridDim.x / 8;
    const size_t last = (blockIdx.x % 8 + 1) * (ntiles / 8);
    for (size_t t = first; t < last; t += step) {
        T v[KPT];
        const T* p = src + t * TILE + threadIdx.x;
#pragma unroll
        for (int i = 0; i < KPT; ++i) v[i] = p[i * WG];
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const unsigned slot = i * WG + threadIdx.x;
            const unsigned d = slot / RUN;
            const size_t o = seq ? t * TILE + slot : d * per_bucket + t * RUN + (slot - d * RUN);
            dst[o + skew] = v[i];
        }
    }
}
template <typename T, int KPT, int WG> void run(size_t bytes, int occ, int skew, int seq) {
    constexpr int TILE = WG * KPT;
    size_t n = bytes / sizeof(T);
    n -= n % ((size_t)TILE * 8);
    T *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes + 4096));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((scat<T, KPT, WG>), dim3(256 * occ), dim3(WG), 0, 0, a, b, n, skew, seq);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((scat<T, KPT, WG>), dim3(256 * occ), dim3(WG), 0, 0, a, b, n, skew, seq);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    printf("elem %2zu B  wg %4d x %2d  run %4zu B  skew %2d occ %d %s: %.3f ms  %.0f GB/s\n", sizeof(T), WG, KPT,
           (size_t)(TILE / 256) * sizeof(T), skew, occ, seq ? "SEQ" : "   ", ms, 2.0 * n * sizeof(T) / ms / 1e6);
    CK(hipFree(a)); CK(hipFree(b));
}
int main() {
    const size_t bytes = 1ull << 30;
    for (int skew : {0, 13}) {
        run<unsigned, 16, 512>(bytes, 2, skew, 0);
        run<unsigned, 16, 512>(bytes, 3, skew, 0);
        run<unsigned, 28, 512>(bytes, 2, skew, 0);
        run<unsigned, 28, 512>(bytes, 2, skew, 1);
        run<unsigned, 32, 512>(bytes, 2, skew, 0);
        run<unsigned, 56, 512>(bytes, 1, skew, 0);
        run<unsigned, 56, 512>(bytes, 2, skew, 0);
        run<unsigned, 28, 1024>(bytes, 1, skew, 0);
        run<unsigned, 28, 1024>(bytes, 2, skew, 0);
        run<unsigned, 64, 1024>(bytes, 1, skew, 0);
        run<unsigned long long, 14, 512>(bytes, 2, skew, 0);
        run<unsigned long long, 28, 512>(bytes, 2, skew, 0);
        run<unsigned long long, 28, 1024>(bytes, 1, skew, 0);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
