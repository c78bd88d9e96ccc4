// Round-trip latency of a flag hand-off between two workgroups, by load/store cache-scope bits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int LM> __device__ __forceinline__ unsigned ld(const unsigned* p) {
    unsigned v;
    if (LM == 0) asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LM == 1) asm volatile("global_load_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LM == 2) asm volatile("global_load_dword %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LM == 3) asm volatile("global_load_dword %0, %1, off nt sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int SM> __device__ __forceinline__ void st(unsigned* p, unsigned v) {
    if (SM == 0) asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    if (SM == 1) asm volatile("global_store_dword %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
    if (SM == 2) asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    if (SM == 3) asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v) : "memory");
}

template <int LM, int SM>
__global__ void pp(unsigned* flags, int a, int b, int rounds, unsigned long long* out) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x != 0) return;
    const int me = blockIdx.x;
    if (me != a && me != b) return;
    unsigned* f0 = flags;        // a -> b
    unsigned* f1 = flags + 64;   // b -> a (another line)
    bool fail = false;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 1; i <= rounds && !fail; ++i) {
        if (me == a) {
            st<SM>(f0, (unsigned)i);
            int spin = 0;
            while (ld<LM>(f1) != (unsigned)i) if (++spin > (1 << 18)) { fail = true; break; }
        } else {
            int spin = 0;
            while (ld<LM>(f0) != (unsigned)i) if (++spin > (1 << 18)) { fail = true; break; }
            st<SM>(f1, (unsigned)i);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const int slot = (me == a) ? 0 : 1;
    out[slot * 4 + 0] = t1 - t0;
    out[slot * 4 + 1] = xcc & 0xf;
    out[slot * 4 + 2] = fail;
}

template <int LM, int SM> void run(const char* name, int a, int b) {
    unsigned* flags; unsigned long long* out;
    CK(hipMalloc(&flags, 1024)); CK(hipMalloc(&out, 64));
    CK(hipMemset(flags, 0, 1024)); CK(hipMemset(out, 0, 64));
    const int rounds = 2000;
    hipLaunchKernelGGL((pp<LM, SM>), dim3(16), dim3(64), 0, 0, flags, a, b, rounds, out);
    CK(hipDeviceSynchronize());
    unsigned long long h[8]; CK(hipMemcpy(h, out, 64, hipMemcpyDeviceToHost));
    printf("%-28s blocks %d,%d xcc %llu,%llu  %7.1f ticks/round-trip%s\n", name, a, b, h[1], h[5], (double)h[0] / rounds,
           (h[2] || h[6]) ? "  FAILED (stale)" : "");
    CK(hipFree(flags)); CK(hipFree(out));
}

int main() {
    for (int pass = 0; pass < 2; ++pass) {
        const int a = 0, b = pass == 0 ? 8 : 1;
        printf("--- %s\n", pass == 0 ? "same XCD expected" : "different XCD expected");
        run<0, 0>("ld sc1 / st sc1", a, b);
        run<1, 0>("ld sc0 / st sc1", a, b);
        run<1, 1>("ld sc0 / st sc0", a, b);
        run<1, 3>("ld sc0 / st plain", a, b);
        run<2, 2>("ld sc0sc1 / st sc0sc1", a, b);
        run<3, 0>("ld nt sc1 / st sc1", a, b);
        run<0, 3>("ld sc1 / st plain", a, b);
    }
    return 0;
}
