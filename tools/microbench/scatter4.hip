// Does a wider per-lane store help the scattered-run write pattern?  Same pattern as scatter_copy.hip
// (u32, 8192-key tiles, 256 runs of 32 elements, runs unaligned by `skew`), three store shapes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
struct __attribute__((packed, aligned(4))) Q { unsigned a, b, c, d; };
struct __attribute__((packed, aligned(4))) D2 { unsigned a, b; };
template <int MODE>
__global__ __launch_bounds__(512) void scat(const unsigned* __restrict__ src, unsigned* __restrict__ dst, size_t n, int skew) {
    constexpr int KPT = 16, TILE = 512 * KPT, RUN = 32;
    const size_t ntiles = n / TILE, per_bucket = n / 256;
    const size_t first = (blockIdx.x % 8) * (ntiles / 8) + blockIdx.x / 8, step = gridDim.x / 8, last = (blockIdx.x % 8 + 1) * (ntiles / 8);
    for (size_t t = first; t < last; t += step) {
        unsigned v[KPT];
        const unsigned* p = src + t * TILE + threadIdx.x;
#pragma unroll
        for (int i = 0; i < KPT; ++i) v[i] = p[i * 512];
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const unsigned slot = i * 512 + threadIdx.x;
                dst[(slot / RUN) * per_bucket + t * RUN + (slot % RUN) + skew] = v[i];
            }
        } else if (MODE == 1) {  // thread owns 4 consecutive slots: one 16-byte store (4-byte aligned)
#pragma unroll
            for (int i = 0; i < KPT; i += 4) {
                const unsigned slot = (i / 4) * 2048 + threadIdx.x * 4;
                Q q{v[i], v[i + 1], v[i + 2], v[i + 3]};
                *reinterpret_cast<Q*>(&dst[(slot / RUN) * per_bucket + t * RUN + (slot % RUN) + skew]) = q;
            }
        } else {  // 2 consecutive slots: 8-byte store
#pragma unroll
            for (int i = 0; i < KPT; i += 2) {
                const unsigned slot = (i / 2) * 1024 + threadIdx.x * 2;
                D2 q{v[i], v[i + 1]};
                *reinterpret_cast<D2*>(&dst[(slot / RUN) * per_bucket + t * RUN + (slot % RUN) + skew]) = q;
            }
        }
    }
}
template <int MODE> void run(const char* name, int skew) {
    const size_t bytes = 1ull << 30, n = bytes / 4;
    unsigned *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes + 4096));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((scat<MODE>), dim3(512), dim3(512), 0, 0, a, b, n, skew);
    CK(hipEventRecord(e0));
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((scat<MODE>), dim3(512), dim3(512), 0, 0, a, b, n, skew);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("%-24s skew %2d: %.3f ms  %.0f GB/s\n", name, skew, ms, 2.0 * bytes / ms / 1e6);
    CK(hipFree(a)); CK(hipFree(b));
}
int main() {
    for (int skew : {0, 13, 16, 1}) {
        run<0>("dword per lane", skew);
        run<2>("dwordx2 per lane", skew);
        run<1>("dwordx4 per lane", skew);
    }
    return 0;
}
