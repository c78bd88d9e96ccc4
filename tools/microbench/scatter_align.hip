// Does it pay to issue the sweep's run writes line by line?  256 runs per tile at pseudo-random bases (no channel
// camping); (a) "slots": thread t writes tile slot i*512+t (a wave instruction = 256 consecutive tile bytes cut by
// run ends and by line boundaries); (b) "lines": a half-wave writes one aligned 128-byte line of one run.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int WG = 512;
template <typename T, int KPT, int MODE>
__global__ __launch_bounds__(WG) void scat(const T* __restrict__ src, T* __restrict__ dst, size_t n, const unsigned* __restrict__ off) {
    constexpr int TILE = WG * KPT;
    constexpr int RUN = TILE / 256;
    constexpr int LINE = 128 / sizeof(T);
    const size_t ntiles = n / TILE;
    const size_t per_bucket = ntiles * RUN + 256;
    const size_t first = (blockIdx.x % 8) * (ntiles / 8) + blockIdx.x / 8;
    const size_t step = gridDim.x / 8;
    const size_t last = (blockIdx.x % 8 + 1) * (ntiles / 8);
    for (size_t t = first; t < last; t += step) {
        T v[KPT];
        const T* p = src + t * TILE + threadIdx.x;
#pragma unroll
        for (int i = 0; i < KPT; ++i) v[i] = p[i * WG];
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const unsigned slot = i * WG + threadIdx.x;
                const unsigned d = slot / RUN;
                dst[d * per_bucket + off[d] + t * RUN + (slot - d * RUN)] = v[i];
            }
        } else {
            T x = 0;
#pragma unroll
            for (int i = 0; i < KPT; ++i) x ^= v[i];
            const unsigned w = threadIdx.x >> 6, lane = threadIdx.x & 63;
            constexpr int HALVES = 64 / LINE;  // runs handled side by side in one wave instruction
            const unsigned h = lane / LINE, l = lane % LINE;
            for (unsigned k = 0; k < 32 / HALVES; ++k) {
                const unsigned d = 32 * w + HALVES * k + h;
                const size_t g = d * per_bucket + off[d] + t * RUN;
                const size_t e = g + RUN;
                for (size_t j = g & ~(size_t)(LINE - 1); j < e; j += LINE) {
                    const size_t a = j + l;
                    if (a >= g && a < e) dst[a] = x;
                }
            }
        }
    }
}
template <typename T, int KPT, int MODE> void run(size_t bytes, int occ, int aligned) {
    constexpr int TILE = WG * KPT;
    size_t n = bytes / sizeof(T);
    n -= n % ((size_t)TILE * 8);
    T *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes + (1 << 20)));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    unsigned h_off[256], *d_off; CK(hipMalloc(&d_off, sizeof h_off));
    unsigned s = 12345;
    for (int i = 0; i < 256; ++i) { s = s * 1664525u + 1013904223u; h_off[i] = aligned ? ((s >> 16) % 8) * 32 : (s >> 16) % 251; }
    CK(hipMemcpy(d_off, h_off, sizeof h_off, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((scat<T, KPT, MODE>), dim3(256 * occ), dim3(WG), 0, 0, a, b, n, d_off);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((scat<T, KPT, MODE>), dim3(256 * occ), dim3(WG), 0, 0, a, b, n, d_off);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    printf("elem %2zu B  512 x %2d  run %4zu B  %s %s occ %d: %.3f ms  %.0f GB/s\n", sizeof(T), KPT, (size_t)(TILE / 256) * sizeof(T),
           MODE ? "lines" : "slots", aligned ? "run bases on lines" : "run bases anywhere", occ, ms, 2.0 * n * sizeof(T) / ms / 1e6);
    CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(d_off));
}
int main() {
    const size_t bytes = 1ull << 30;
    for (int al : {0, 1}) {
        run<unsigned, 16, 0>(bytes, 2, al);
        run<unsigned, 16, 1>(bytes, 2, al);
        run<unsigned, 28, 0>(bytes, 2, al);
        run<unsigned, 28, 1>(bytes, 2, al);
        run<unsigned, 32, 0>(bytes, 2, al);
        run<unsigned, 32, 1>(bytes, 2, al);
        run<unsigned long long, 14, 0>(bytes, 2, al);
        run<unsigned long long, 14, 1>(bytes, 2, al);
        run<unsigned long long, 28, 0>(bytes, 2, al);
        run<unsigned long long, 28, 1>(bytes, 2, al);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
