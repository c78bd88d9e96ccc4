#include <hip/hip_runtime.h>
#include <cstdio>
// cycles per wave64 VALU instruction on one SIMD, by waves per SIMD
template <int KIND>
__global__ void k(unsigned* out, int iters) {
    unsigned a = threadIdx.x, b = a * 3 + 1, c = a ^ 0x55, d = a + 7, e = b + 11, f = c + 13, g = d + 17, h = e + 19;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {  // v_and/xor chain (8 independent)
            asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %4\n"
                         "v_xor_b32 %4, %4, %5\n v_xor_b32 %5, %5, %6\n v_xor_b32 %6, %6, %7\n v_xor_b32 %7, %7, %0"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
        } else if (KIND == 1) {  // bitop3
            asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x90\n v_bitop3_b32 %1, %1, %2, %3 bitop3:0x90\n"
                         "v_bitop3_b32 %2, %2, %3, %4 bitop3:0x90\n v_bitop3_b32 %3, %3, %4, %5 bitop3:0x90\n"
                         "v_bitop3_b32 %4, %4, %5, %6 bitop3:0x90\n v_bitop3_b32 %5, %5, %6, %7 bitop3:0x90\n"
                         "v_bitop3_b32 %6, %6, %7, %0 bitop3:0x90\n v_bitop3_b32 %7, %7, %0, %1 bitop3:0x90"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
        } else if (KIND == 2) {  // v_cmp to sgpr pairs
            asm volatile("v_cmp_ne_u32_e64 s[90:91], 0, %0\n v_cmp_ne_u32_e64 s[92:93], 0, %1\n"
                         "v_cmp_ne_u32_e64 s[94:95], 0, %2\n v_cmp_ne_u32_e64 s[96:97], 0, %3\n"
                         "v_cmp_ne_u32_e64 s[90:91], 0, %4\n v_cmp_ne_u32_e64 s[92:93], 0, %5\n"
                         "v_cmp_ne_u32_e64 s[94:95], 0, %6\n v_cmp_ne_u32_e64 s[96:97], 0, %7"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)
                         :: "s90","s91","s92","s93","s94","s95","s96","s97");
        } else if (KIND == 3) {  // bfe_i32
            asm volatile("v_bfe_i32 %0, %1, 3, 1\n v_bfe_i32 %1, %2, 3, 1\n v_bfe_i32 %2, %3, 3, 1\n v_bfe_i32 %3, %4, 3, 1\n"
                         "v_bfe_i32 %4, %5, 3, 1\n v_bfe_i32 %5, %6, 3, 1\n v_bfe_i32 %6, %7, 3, 1\n v_bfe_i32 %7, %0, 3, 1"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
        } else {  // the real match block shape: bfe, cmp, 2 bitop3 with sgpr
            asm volatile("v_bfe_i32 %2, %0, 0, 1\n v_cmp_ne_u32_e64 s[90:91], 0, %2\n v_bfe_i32 %3, %0, 1, 1\n v_cmp_ne_u32_e64 s[92:93], 0, %3\n"
                         "v_bitop3_b32 %4, %4, s90, %2 bitop3:0x90\n v_bitop3_b32 %5, %5, s91, %2 bitop3:0x90\n"
                         "v_bitop3_b32 %4, %4, s92, %3 bitop3:0x90\n v_bitop3_b32 %5, %5, s93, %3 bitop3:0x90"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)
                         :: "s90","s91","s92","s93");
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (unsigned)(t1 - t0); }
    out[1 + threadIdx.x % 8] = a + b + c + d + e + f + g + h;
}
template <int KIND> void run(const char* name) {
    unsigned* out; hipMalloc(&out, 4096);
    const int iters = 20000;
    for (int wps : {1, 2, 4, 8}) {  // waves per SIMD: block of wps*4 waves, 1 block per CU (256 blocks)
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(wps * 256), 0, 0, out, iters);
        hipDeviceSynchronize();
        unsigned cyc; hipMemcpy(&cyc, out, 4, hipMemcpyDeviceToHost);
        printf("%-10s waves/SIMD %d: %.2f cycles per instr per wave, SIMD throughput %.2f cycles/instr\n", name, wps,
               (double)cyc / (iters * 8.0), (double)cyc / (iters * 8.0) / wps);
    }
    hipFree(out);
}
int main() { run<0>("v_xor"); run<1>("v_bitop3"); run<2>("v_cmp_sgpr"); run<3>("v_bfe_i32"); run<4>("match-mix"); return 0; }
