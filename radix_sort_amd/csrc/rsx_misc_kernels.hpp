// rsx_misc_kernels.hpp -- the kernels that do not depend on the element size: digit totals, the
// one-byte counting path, the two device self-tests, the multi-GPU splitter search and the harness
// (generators, verifier).  Included by rsx.hip only (one definition per library).
#pragma once
#include "rsx_device.hpp"

namespace rsx {

// The 256 digit totals of a count matrix (column sums over the regions): what the per-pass
// building blocks hand to the multi-GPU driver, and what the one-byte counting path expands.
// The prefix phase proper (mod.rs:110-120) runs inside the sweep kernel's prologue.
__global__ __launch_bounds__(256) void rsx_totals_kernel(const unsigned long long* __restrict__ J,
                                                         uint32_t num_regions, uint64_t* __restrict__ counts_out, uint32_t j32) {
    const uint32_t tid = threadIdx.x;
    uint64_t c = 0;
    for (uint32_t r0 = 0; r0 < (uint32_t)J_REPL * num_regions; r0 += 8) {  // all replicas, all regions; 8 loads in flight
        uint64_t part[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k)
            part[k] = j32 ? (uint64_t)reinterpret_cast<const uint32_t*>(J)[(r0 + k) * RADIX + tid] : (uint64_t)J[(r0 + k) * RADIX + tid];
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) c += part[k];
    }
    counts_out[tid] = c;
}

// Zeroes n16 x 16 bytes (a control block, by sorts that are being captured into a graph: a 788 KiB memset NODE aborted
// at replay on ROCm 7.2, a kernel node does not).
__global__ __launch_bounds__(256) void rsx_zero16_kernel(uint4* __restrict__ p, uint64_t n16) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x)
        p[i] = make_uint4(0, 0, 0, 0);
}

// ------------------------------------------------------------ one-byte elements --
// An element that IS its one-byte key (u8, i8) is fully described by its digit: the sorted array is
// the 256 counts written out as runs (counting sort: one read of the data, one write, no scatter).
// `counts` are by mapped digit (the count kernel maps signed keys); byte = mapped value ^ xor_mask.
__global__ __launch_bounds__(256) void rsx_expand_bytes_kernel(uint8_t* __restrict__ dst, uint64_t n,
                                                               const unsigned long long* __restrict__ J, uint32_t num_regions,
                                                               uint32_t j32, uint32_t xor_mask) {
    __shared__ uint64_t start[RADIX + 1];
    __shared__ uint64_t wsum[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the 256 digit totals of the count matrix (all replicas, all regions): every block sums them itself
    // (L2-resident, 64-128 independent loads per thread) -- no totals launch in between
    uint64_t c = 0;
    {
        const uint32_t rows = (uint32_t)J_REPL * num_regions;
        for (uint32_t r0 = 0; r0 < rows; r0 += 8) {
            uint64_t part[8];
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) {
                const uint32_t r = r0 + k;  // rows is a multiple of 8 (J_REPL == 8)
                part[k] = j32 ? (uint64_t)reinterpret_cast<const uint32_t*>(J)[r * RADIX + tid] : (uint64_t)J[r * RADIX + tid];
            }
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) c += part[k];
        }
    }
    uint64_t x = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t y = __shfl_up(x, o);
        if (lane >= (uint32_t)o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    uint64_t wb = 0;
    for (uint32_t w = 0; w < wave; ++w) wb += wsum[w];
    start[tid] = wb + x - c;
    if (tid == RADIX - 1) start[RADIX] = wb + x;
    __syncthreads();
    const bool wide = (reinterpret_cast<uintptr_t>(dst) & 15u) == 0;  // 16-byte stores need the alignment
    const uint64_t chunks = (n + 15) / 16;
    // last v with start[v] <= p
    auto digit_at = [&](uint64_t p) {
        uint32_t lo = 0, hi = RADIX;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) / 2;
            if (start[mid] <= p) lo = mid;
            else hi = mid;
        }
        return lo;
    };
    // a wave writes 64 consecutive chunks (1 KiB) per step; a run is n/256 bytes on average, so nearly every
    // step lies inside ONE run: one search by the wave (all lanes read the same words: broadcasts), then splat
    for (uint64_t base = ((uint64_t)blockIdx.x * 4 + wave) * 64; base < chunks; base += (uint64_t)gridDim.x * 4 * 64) {
        const uint64_t ch = base + lane;
        const uint64_t span0 = base * 16;
        const uint32_t v0 = digit_at(span0);  // wave-uniform
        uint32_t w[4];
        const uint64_t p0 = ch * 16;
        if (start[v0 + 1] >= span0 + 64 * 16) {  // wave-uniform: the whole span is digit v0
            const uint32_t byte = (v0 ^ xor_mask) & 0xFFu;
            w[0] = w[1] = w[2] = w[3] = byte * 0x01010101u;
        } else {
            uint32_t v = digit_at(p0);
            w[0] = w[1] = w[2] = w[3] = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                while (v < RADIX - 1 && start[v + 1] <= p0 + k) ++v;  // empty digits are stepped over
                w[k / 4] |= ((v ^ xor_mask) & 0xFFu) << (8 * (k % 4));
            }
        }
        if (ch < chunks) {
            if (wide && p0 + 16 <= n) {
                *reinterpret_cast<uint4*>(dst + p0) = make_uint4(w[0], w[1], w[2], w[3]);
            } else {
                for (int k = 0; k < 16 && p0 + k < n; ++k) dst[p0 + k] = (uint8_t)(w[k / 4] >> (8 * (k % 4)));
            }
        }
    }
}

// ------------------------------------------------------------ two-byte elements --
// An element that IS its two-byte key (u16, i16) is fully described by its 65536-bin histogram: count, then write the
// runs (one read and one write of the data instead of D = 2 passes of each; same bytes as mod.rs:84-169 leaves).
//
// rsx_count16_kernel: one workgroup per CU keeps all 65536 counters in LDS as 16-bit halves of 32768 words (128 KiB) and
// counts its contiguous share of the input with RETURNED atomics: the lane whose add takes a counter from 0x7FFF to
// 0x8000 moves 0x8000 of it to a global overflow table (one lane per 32768 increments of a bin sees that value, and a
// counter cannot run from 0x8000 to 0x10000 before that lane's subtraction lands: at most a few thousand adds are
// in flight), so no counter ever carries into its neighbour, whatever the skew.  At the end the workgroup stores its
// counters: P[workgroup][32768].  bin = key ^ xor_mask (0x8000 for signed keys: radix_digits.rs:55-69).
__global__ __launch_bounds__(1024) void rsx_count16_kernel(const uint16_t* __restrict__ src, uint64_t n, uint32_t xor_mask,
                                                           uint32_t* __restrict__ P, uint32_t* __restrict__ ovf) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    uint32_t* cnt = reinterpret_cast<uint32_t*>(smem16);  // [32768]
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < 32768u / 4u; i += 1024u) reinterpret_cast<uint4*>(cnt)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    auto park = [&](uint32_t bin, uint32_t sh) {  // the add that took a counter to (or over) 0x8000 moves that half to the overflow table
        atomicSub(&cnt[bin >> 1], 0x8000u << sh);
        atomicAdd(&ovf[bin], 0x8000u);
    };
    auto count = [&](uint32_t key) {
        const uint32_t bin = (key ^ xor_mask) & 0xFFFFu;
        const uint32_t sh = (bin & 1u) * 16u;
        const uint32_t old = atomicAdd(&cnt[bin >> 1], 1u << sh);
        if (((old >> sh) & 0xFFFFu) == 0x7FFFu) park(bin, sh);
    };
    auto count_wave = [&](uint32_t key, uint32_t c) {  // one lane adds for `c` lanes that hold the same key
        const uint32_t bin = (key ^ xor_mask) & 0xFFFFu;
        const uint32_t sh = (bin & 1u) * 16u;
        const uint32_t before = (atomicAdd(&cnt[bin >> 1], c << sh) >> sh) & 0xFFFFu;
        if (before < 0x8000u && before + c >= 0x8000u) park(bin, sh);
    };
    // the elements ahead of the first 16-byte boundary and behind the last whole 16-byte pack: workgroup 0, one by one
    uint64_t head = ((16u - (uint32_t)(reinterpret_cast<uintptr_t>(src) & 15u)) & 15u) / 2u;
    if (head > n) head = n;
    const uint64_t npack = (n - head) / 8;
    if (blockIdx.x == 0) {
        if (tid < head) count(src[tid]);
        const uint64_t t0 = head + npack * 8;
        if (tid >= 64 && t0 + (tid - 64) < n) count(src[t0 + (tid - 64)]);
    }
    const uint64_t per = (npack + gridDim.x - 1) / gridDim.x;
    const uint64_t p0 = (uint64_t)blockIdx.x * per;
    uint64_t p1 = p0 + per;
    if (p1 > npack) p1 = npack;
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    const v4u* vsrc = reinterpret_cast<const v4u*>(src + head);
    for (uint64_t i = p0 + tid; i < p1; i += 1024) {
        const v4u v = __builtin_nontemporal_load(vsrc + i);
        // skewed inputs put whole waves on one key, and same-address atomics of one instruction serialise (a constant
        // array: 874 us per 2^28 keys): when every lane holds the first lane's 16 bytes, one lane adds for the wave
        const uint32_t f0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)v[0]), f1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)v[1]);
        const uint32_t f2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)v[2]), f3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)v[3]);
        const uint64_t same = __ballot(((v[0] ^ f0) | (v[1] ^ f1) | (v[2] ^ f2) | (v[3] ^ f3)) == 0u);
        if (same == __builtin_amdgcn_read_exec()) {
            if (mbcnt64(same) == 0) {
                const uint32_t c = (uint32_t)__popcll(same);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    count_wave(v[k] & 0xFFFFu, c);
                    count_wave(v[k] >> 16, c);
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                count(v[k] & 0xFFFFu);
                count(v[k] >> 16);
            }
        }
    }
    __syncthreads();
    uint4* out = reinterpret_cast<uint4*>(P + (size_t)blockIdx.x * 32768u);
    for (uint32_t i = tid; i < 32768u / 4u; i += 1024u) out[i] = reinterpret_cast<const uint4*>(cnt)[i];
}

// Bin totals: grid = 256 (bins [256 j, 256 j + 256)), block = 256.  tot[bin] = overflow + sum over the count
// kernel's workgroups; BT[j] = the block's sum.  The overflow entries are put back to zero (clean for the next sort).
__global__ __launch_bounds__(256) void rsx_total16_kernel(const uint32_t* __restrict__ P, uint32_t parts, uint32_t* __restrict__ ovf,
                                                          uint64_t* __restrict__ tot, uint64_t* __restrict__ BT) {
    __shared__ uint64_t ws[4];
    const uint32_t tid = threadIdx.x, bin = blockIdx.x * 256u + tid;
    const uint32_t w = bin >> 1, sh = (bin & 1u) * 16u;
    uint64_t c = ovf[bin];
    if (c) ovf[bin] = 0;
    constexpr uint32_t FLY = 32;  // loads in flight per thread (512 bytes per workgroup and part, 128 KiB apart: with 8 the 256
                                  // parts of a large array were 32 round trips, 18 us)
    for (uint32_t b0 = 0; b0 < parts; b0 += FLY) {
        uint32_t part[FLY];
#pragma unroll
        for (uint32_t k = 0; k < FLY; ++k) part[k] = b0 + k < parts ? P[(size_t)(b0 + k) * 32768u + w] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < FLY; ++k) c += (part[k] >> sh) & 0xFFFFu;
    }
    tot[bin] = c;
    uint64_t x = c;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    if ((tid & 63u) == 0u) ws[tid >> 6] = x;
    __syncthreads();
    if (tid == 0) BT[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// Exclusive scan of the 65536 bin totals -> starts[65537], and the verdict of the wide-key hybrid.  grid = 256 workgroups
// of 256 threads: workgroup b scans bins [256 b, 256 b + 256) from the block totals BT (rsx_total16_kernel) of the blocks
// before it; all add what they see to WidePlan::scan_cnt -- buckets above what a workgroup of 256 / 512 / 1024 threads
// holds, groups of 2^g consecutive buckets (g = 2 .. 6, from the butterfly over a wave's 64 bins) above what 512 hold --
// and the last one to finish decides.  (As ONE workgroup this took 49-76 us in three forms: a single CU moves its 1.5 MB
// no faster; 12 % of a 2^23-key sort.)
__global__ __launch_bounds__(256) void rsx_scan16_kernel(const uint64_t* __restrict__ tot, const uint64_t* __restrict__ BT,
                                                         uint64_t* __restrict__ starts, uint64_t cap256, uint64_t cap512, uint64_t cap1024,
                                                         uint32_t gs_max, uint32_t forced, uint64_t medium_max, uint64_t crowd_max,
                                                         WidePlan* __restrict__ plan, uint32_t* __restrict__ host_verdict) {
    __shared__ uint64_t ws[4], wb[4], s_big[4], s_crowd[4];
    __shared__ uint32_t s_cnt[4][8];
    __shared__ uint32_t s_last;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, b = blockIdx.x;
    const uint64_t c = tot[(size_t)b * 256u + tid];
    uint64_t before = tid < b ? BT[tid] : 0ull;  // blocks before mine
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
    uint64_t x = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t y = __shfl_up(x, o);
        if (lane >= (uint32_t)o) x += y;
    }
    if (lane == 63) ws[wave] = x;
    if (lane == 0) wb[wave] = before;
    // what this wave's 64 bins add to the counts
    const uint32_t o256 = (uint32_t)__popcll(__ballot(c > cap256)), o512 = (uint32_t)__popcll(__ballot(c > cap512)),
                   o1024 = (uint32_t)__popcll(__ballot(c > cap1024));
    uint32_t og[7] = {0, 0, 0, 0, 0, 0, 0};
    uint64_t sum = c;
#pragma unroll
    for (int g = 1; g <= 6; ++g) {
        sum += __shfl_xor(sum, 1 << (g - 1));  // every lane: the sum of its aligned group of 2^g bins
        if (g >= 2) og[g] = (uint32_t)__popcll(__ballot((lane & ((1u << g) - 1u)) == 0u && sum > cap512));
    }
    uint64_t big = c, crowd = c > cap1024 ? c : 0ull;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t y = __shfl_xor(big, o);
        big = y > big ? y : big;
        crowd += __shfl_xor(crowd, o);
    }
    // (one set of atomics per WORKGROUP, every word on a line of its own: per wave and on one cache line of the plan -- which
    // the memory side serialises at ~12 ns apiece: 1024 waves x up to 10 of them -- this kernel took 33 us at 2^23 keys and
    // 92 us at 2^30, where every bucket exceeds the two smaller forms)
    if (lane == 0) {
        s_cnt[wave][0] = o256;
        s_cnt[wave][1] = o512;
        s_cnt[wave][2] = o1024;
#pragma unroll
        for (int g = 2; g <= 6; ++g) s_cnt[wave][1 + g] = og[g];
        s_big[wave] = big;
        s_crowd[wave] = crowd;
    }
    __syncthreads();
    if (tid < 8) {
        const uint32_t v = s_cnt[0][tid] + s_cnt[1][tid] + s_cnt[2][tid] + s_cnt[3][tid];
        if (v) atomicAdd(&plan->scan_cnt[tid * WidePlan::SCAN_LINE], v);
    } else if (tid == 8) {
        uint64_t m = s_big[0];
        for (int w = 1; w < 4; ++w) m = s_big[w] > m ? s_big[w] : m;
        atomicMax(&plan->scan_max, m > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)m);
    } else if (tid == 9) {
        const uint64_t cr = s_crowd[0] + s_crowd[1] + s_crowd[2] + s_crowd[3];
        if (cr) atomicAdd(&plan->scan_big, (unsigned long long)cr);
    }
    uint64_t run = wb[0] + wb[1] + wb[2] + wb[3] + x - c;
    for (uint32_t w = 0; w < wave; ++w) run += ws[w];
    starts[(size_t)b * 256u + tid] = run;
    if (b == gridDim.x - 1 && tid == 255) starts[65536] = run + c;
    // the last workgroup to get here decides
    __threadfence();
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(&plan->scan_done, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (s_last == 0 || tid != 0) return;
    __threadfence();
    uint32_t cnt[8];
    for (int i = 0; i < 8; ++i) cnt[i] = __hip_atomic_load(&plan->scan_cnt[i * WidePlan::SCAN_LINE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t n256 = cnt[0], n512 = cnt[1], n1024 = cnt[2];
    // The smallest workgroup that holds all but a handful of the buckets (those few go through memory, one workgroup
    // each: tolerable for buckets of its own size class, not for what exceeds the largest workgroup -- then the LSD
    // passes run, unless the hybrid is forced).  Small buckets in groups if the host's average says so and no bucket
    // is larger than a group's workgroup: the largest group size (up to the host's) whose groups fit.
    constexpr uint32_t FEW = 8;
    const uint32_t violation = __hip_atomic_load(&plan->violation, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (set by an earlier kernel of the stream)
    uint32_t gshift = 0;
    for (uint32_t g = gs_max <= 6u ? gs_max : 6u; g >= 2u && gshift == 0u; --g)
        if (cnt[1 + g] <= FEW) gshift = g;
    plan->group_shift = gshift;
    // Buckets above the chosen form's workgroup go to rsx_bucket16_medium_kernel (one pass through memory that splits
    // them by their next bits, then LDS; one workgroup each, a tenth of the other kernel's rate per element): tolerable
    // for a handful, the largest at most `medium_max` (eight more bits bring it down to a workgroup's size), together at
    // most `crowd_max` elements (n / 64), else -- unless the hybrid is forced -- the LSD passes run.
    const uint32_t biggest = __hip_atomic_load(&plan->scan_max, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t v;
    const uint64_t crowded = __hip_atomic_load(&plan->scan_big, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (violation != 0 || (!forced && (n1024 > FEW || (n1024 != 0 && ((uint64_t)biggest > medium_max || crowded > crowd_max))))) v = VERDICT_LSD;
    else if (gshift != 0 && n512 == 0) v = VERDICT_HYBRID | VERDICT_GROUPS;
    else if (n256 <= FEW) v = VERDICT_HYBRID | VERDICT_WG256 | (n256 ? VERDICT_MEDIUM : 0u);
    else if (n512 <= FEW) v = VERDICT_HYBRID | VERDICT_WG512 | (n512 ? VERDICT_MEDIUM : 0u);
    else v = VERDICT_HYBRID | VERDICT_WG1024 | (n1024 ? VERDICT_MEDIUM : 0u);
    plan->verdict = v;
    __hip_atomic_store(host_verdict, (v & VERDICT_HYBRID) ? 1u : 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // the host's forecast for later sorts
}

// Writes the runs.  The output is cut into 1 KiB steps (64 lanes x 8 elements); a wave takes every 4th step of its
// workgroup's contiguous share.  Two-level search: the 256 bin-block starts (scan of BT, per workgroup, in LDS) and
// the 256 bin starts inside the block at hand (scan of its tot[], per wave, in LDS, reloaded when the wave moves on).
// A step inside ONE bin -- n / 65536 elements on average -- is one search and a splat.
__global__ __launch_bounds__(256) void rsx_expand16_kernel(uint16_t* __restrict__ dst, uint64_t n, const uint64_t* __restrict__ tot,
                                                           const uint64_t* __restrict__ BT, uint32_t xor_mask) {
    __shared__ uint64_t bt[257];        // exclusive scan of BT
    __shared__ uint64_t st[4][257];     // per wave: bin starts of its current bin-block (absolute positions)
    __shared__ uint64_t wsum[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    {
        const uint64_t c = BT[tid];
        uint64_t x = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t y = __shfl_up(x, o);
            if (lane >= (uint32_t)o) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        uint64_t wb = 0;
        for (uint32_t k = 0; k < wave; ++k) wb += wsum[k];
        bt[tid] = wb + x - c;
        if (tid == 255) bt[256] = wb + x;
        __syncthreads();
    }
    const bool wide = (reinterpret_cast<uintptr_t>(dst) & 15u) == 0;
    const uint64_t steps = (n + 511) / 512;
    const uint64_t per = (steps + gridDim.x - 1) / gridDim.x;
    const uint64_t s0 = (uint64_t)blockIdx.x * per;
    uint64_t s1 = s0 + per;
    if (s1 > steps) s1 = steps;
    uint64_t* my = st[wave];
    uint32_t cur = ~0u;  // bin-block whose starts `my` holds (wave-uniform)
    auto load_block = [&](uint32_t j) {  // wave-uniform j
        uint64_t c[4], run = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k] = tot[(size_t)j * 256u + lane * 4u + k];
        const uint64_t mine = c[0] + c[1] + c[2] + c[3];
        uint64_t x = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t y = __shfl_up(x, o);
            if (lane >= (uint32_t)o) x += y;
        }
        run = bt[j] + x - mine;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            my[lane * 4u + k] = run;
            run += c[k];
        }
        if (lane == 63) my[256] = run;
        cur = j;
    };
    auto upper = [&](const uint64_t* a, uint64_t p) -> uint32_t {  // last i in [0, 256) with a[i] <= p  (a[0] <= p given)
        uint32_t lo = 0, hi = 256;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (a[mid] <= p) lo = mid;
            else hi = mid;
        }
        return lo;
    };
    uint32_t jb = 0, bb = 0;  // bin-block / bin inside it that hold the wave's position: they only move forward
    for (uint64_t s = s0 + wave; s < s1; s += 4) {
        const uint64_t p0 = s * 512;                      // wave-uniform
        const uint64_t pend = p0 + 512 < n ? p0 + 512 : n;
        const uint64_t q = p0 + (uint64_t)lane * 8;       // this lane's 8 elements
        uint32_t w[4] = {0, 0, 0, 0};
        uint64_t done = p0;                               // positions below `done` are filled in (wave-uniform)
        uint32_t guard = 0;
        while (done < pend && ++guard <= 600u) {          // (a step holds at most 512 bins: bounded whatever the tables say)
            if (cur == ~0u) {                             // first step of the wave: search; afterwards walk
                jb = upper(bt, done);
            } else {
                while (jb < 255u && bt[jb + 1] <= done) ++jb;  // empty blocks are stepped over
            }
            if (jb != cur) {
                load_block(jb);
                bb = upper(my, done);
            } else {
                while (bb < 255u && my[bb + 1] <= done) ++bb;
            }
            const uint64_t lim = my[256] < pend ? my[256] : pend;  // end of what this block covers of the step
            if (my[bb + 1] >= lim) {                      // one bin up to `lim`: splat
                const uint32_t key = ((jb * 256u + bb) ^ xor_mask) & 0xFFFFu;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (q + k >= done && q + k < lim) w[k >> 1] |= key << (16 * (k & 1));
            } else if (q + 8 > done && q < lim) {         // lanes with elements in [done, lim): one search, then walk
                const uint64_t first = q > done ? q : done;
                uint32_t bk = upper(my, first);
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (q + k >= done && q + k < lim) {
                        while (bk < 255u && my[bk + 1] <= q + k) ++bk;
                        const uint32_t key = ((jb * 256u + bk) ^ xor_mask) & 0xFFFFu;
                        w[k >> 1] |= key << (16 * (k & 1));
                    }
            }
            done = lim;
        }
        if (q < n) {
            if (wide && q + 8 <= n) {
                *reinterpret_cast<uint4*>(dst + q) = make_uint4(w[0], w[1], w[2], w[3]);
            } else {
                for (int k = 0; k < 8 && q + k < n; ++k) dst[q + k] = (uint16_t)(w[k >> 1] >> (16 * (k & 1)));
            }
        }
    }
}

// ------------------------------------------------------------------ harness --
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ uint64_t rand64(uint64_t seed, uint64_t index) {
    return splitmix64(seed + index * 0x9E3779B97F4A7C15ull);
}

// ------------------------------------------------------------- LDS atomic order --
// Self-test behind a.rank_atomic: when several lanes of ONE ds_add_rtn instruction hit the same
// address, the sweep needs them applied in ascending lane order (then the returned values are
// stable ranks).  The ISA documents no order, so it is established on the device at hand: 64
// address patterns (1..256 distinct addresses; hashed, lane-cyclic, blocked and same-bank layouts;
// both the 32-bit and the packed 16-bit counter forms), every lane checks its two returned values
// against the ballot-derived rank.  Any mismatch sets *fail and the sweep keeps to ballots.
__global__ __launch_bounds__(512) void rsx_lds_order_kernel(uint32_t* __restrict__ fail) {
    __shared__ uint32_t cnt[8][RADIX];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    static constexpr uint32_t BINS[8] = {1, 2, 3, 5, 16, 64, 256, 256};
    uint32_t bad = 0;
    for (uint32_t p = 0; p < 64; ++p) {
        for (uint32_t i = lane; i < RADIX; i += WAVE) cnt[wave][i] = 0;  // LDS serves one wave in order
        const uint32_t bins = BINS[p & 7];
        const uint32_t h = (uint32_t)(splitmix64(((uint64_t)blockIdx.x << 32) ^ (p << 16) ^ (wave << 8) ^ lane) >> 32);
        uint32_t d;
        switch ((p >> 3) & 3) {
            case 0: d = h % bins; break;                        // hashed
            case 1: d = lane % bins; break;                     // cyclic: neighbours differ
            case 2: d = (lane * bins) / WAVE; break;            // blocked: neighbours share
            default: d = ((h % bins) * 32u) % RADIX; break;     // distinct addresses in one bank
        }
        const uint64_t m = match_digit<8>(d);
        const uint32_t below = mbcnt64(m), group = (uint32_t)__popcll(m);
        uint32_t r1, r2;
        if (p & 32) {  // packed: two 16-bit counters per word
            const uint32_t sh = (d & 1u) * 16u;
            r1 = (atomicAdd(&cnt[wave][d >> 1], 1u << sh) >> sh) & 0xFFFFu;
            r2 = (atomicAdd(&cnt[wave][d >> 1], 1u << sh) >> sh) & 0xFFFFu;
        } else {
            r1 = atomicAdd(&cnt[wave][d], 1u);
            r2 = atomicAdd(&cnt[wave][d], 1u);
        }
        if (r1 != below || r2 != group + below) bad = 1;
    }
    if (__ballot(bad != 0) != 0 && lane == 0) atomicOr(fail, 1u);
}

// ------------------------------------------------------------- same-XCD hand-off --
// Self-test behind the L2-resident status words of a verified single-XCD chain (sweep kernel,
// `local_chain`): is a WAVEFRONT-scope store (plain global_store: the line stays dirty in the XCD's
// L2, nothing is written through) made on one CU seen by an AGENT-scope load (sc1: bypasses the
// reader's L1, served by the L2) issued on ANOTHER CU of the same XCD, with no fence in between?
// Blocks b and b + 8 are dealt to one XCD by the dispatcher as observed; the pair checks that itself
// (HW_REG_XCC_ID) and whether it sits on two CUs (HW_REG_HW_ID), then plays ping-pong for `rounds`
// values: A stores k (wavefront scope) and waits for the acknowledgement, B polls for k (agent
// scope) and acknowledges (agent scope).  Every wait is bounded.
// out[0] += pairs that ran the test on one XCD, out[1] += of those on two different CUs,
// out[2] += pairs where a value did not arrive in time (the verdict: must stay 0).
// words: per pair 4 x u32 {data, ack, meta_a, meta_b}, zeroed by the caller.
__global__ __launch_bounds__(64) void rsx_l2_probe_kernel(uint32_t* __restrict__ words, uint32_t* __restrict__ out,
                                                         uint32_t rounds) {
    if (threadIdx.x != 0) return;
    const uint32_t b = blockIdx.x;
    const bool is_a = (b & 8u) == 0u;
    const uint32_t pair = (b >> 4) * 8u + (b & 7u);
    uint32_t* w = words + pair * 4u;
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const uint32_t me = 0x80000000u | ((xcc & 0xFu) << 16) | ((hw >> 8) & 0xFFu);  // CU / SH / SE id bits of HW_ID
    __hip_atomic_store(&w[is_a ? 2 : 3], me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t other = 0;
    for (uint32_t spin = 0; spin < (1u << 18) && other == 0; ++spin) {
        other = __hip_atomic_load(&w[is_a ? 3 : 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (other == 0) __builtin_amdgcn_s_sleep(2);
    }
    if (other == 0 || ((other ^ me) & 0x000F0000u) != 0) return;  // partner not running, or another XCD: no verdict from this pair
    if (is_a) {
        atomicAdd(&out[0], 1u);
        if (((other ^ me) & 0xFFu) != 0) atomicAdd(&out[1], 1u);
    }
    for (uint32_t k = 1; k <= rounds; ++k) {
        if (is_a) {
            __hip_atomic_store(&w[0], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            uint32_t got = 0;
            for (uint32_t spin = 0; spin < (1u << 16) && got != k; ++spin) {
                got = __hip_atomic_load(&w[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (got != k) __builtin_amdgcn_s_sleep(1);
            }
            if (got != k) return;  // B reports the failure
        } else {
            uint32_t got = 0;
            for (uint32_t spin = 0; spin < (1u << 15) && got != k; ++spin) {
                got = __hip_atomic_load(&w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (got != k) __builtin_amdgcn_s_sleep(1);
            }
            if (got != k) {
                atomicAdd(&out[2], 1u);
                return;
            }
            __hip_atomic_store(&w[1], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- generators (harness; shapes of the reference's src/distr.rs) -------------------------------
// Integer arithmetic only, so that the same (seed, index) gives the same key on the device and in
// the CPU restatement the tests hold against them: no libm call whose last bit could differ.
// 2^(2^-i) as 1.63 fixed point, i = 1..32
__device__ static const uint64_t GEN_EXP2_TAB[32] = {
    0xB504F333F9DE6484ull, 0x9837F0518DB8A96Full, 0x8B95C1E3EA8BD6E7ull, 0x85AAC367CC487B15ull,
    0x82CD8698AC2BA1D7ull, 0x8164D1F3BC030773ull, 0x80B1ED4FD999AB6Cull, 0x8058D7D2D5E5F6B1ull,
    0x802C6436D0E04F51ull, 0x8016302F17467628ull, 0x800B179C82028FD1ull, 0x80058BAF7FEE3B5Dull,
    0x8002C5D00FDCFCB7ull, 0x800162E61BED4A49ull, 0x8000B17292F702A4ull, 0x800058B92ABBAE02ull,
    0x80002C5C8DADE4D7ull, 0x8000162E44EAF636ull, 0x80000B1721FA7C19ull, 0x8000058B90DE7E4Dull,
    0x800002C5C8678F37ull, 0x80000162E431DBA0ull, 0x800000B1721872D1ull, 0x80000058B90C1AA9ull,
    0x8000002C5C8605A4ull, 0x800000162E4300E6ull, 0x8000000B17217FF8ull, 0x800000058B90BFDDull,
    0x80000002C5C85FE7ull, 0x8000000162E42FF2ull, 0x80000000B17217F8ull, 0x8000000058B90BFCull};

// floor(2^t) for t = e + f / 2^32, 0 <= e <= 63: the mantissa 2^(f/2^32) is the product of the table
// entries of f's set bits, each product truncated to 1.63 fixed point
__device__ __forceinline__ uint64_t gen_exp2_floor(uint32_t e, uint32_t f) {
    uint64_t m = 1ull << 63;
    for (int i = 0; i < 32; ++i)
        if (f & (0x80000000u >> i)) m = __umul64hi(m, GEN_EXP2_TAB[i]) << 1;
    return m >> (63u - e);
}
// -log2(w / 2^64) for w >= 1 as 32.32 fixed point: exponent from the leading zeros, 32 fraction bits
// by repeated squaring of the 1.63 mantissa
__device__ __forceinline__ uint64_t gen_neg_log2(uint64_t w) {
    const uint32_t lz = (uint32_t)__builtin_clzll(w);
    uint64_t m = w << lz;
    uint32_t frac = 0;
    for (int i = 0; i < 32; ++i) {
        m = __umul64hi(m, m);  // 2.62 fixed point, in [1, 4)
        if (m >> 63) frac |= 0x80000000u >> i;  // >= 2: this bit of the logarithm is set, m / 2 is back in 1.63
        else m <<= 1;
    }
    return (64ull << 32) - ((((uint64_t)(63u - lz)) << 32) | frac);
}

// key (as up to 128 bits lo/hi) for generator `gen`.  iparam: generator-specific integer form of the
// caller's `param`, made once on the host (rsx_generate_device).
__device__ __forceinline__ void gen_key(int gen, uint64_t seed, double param, uint64_t iparam, uint64_t gi,
                                        uint64_t n_total, uint32_t key_bytes, uint64_t& lo, uint64_t& hi) {
    const uint32_t bits = key_bytes * 8;
    lo = hi = 0;
    switch (gen) {
        case 0:  // uniform (distr.rs:40-52 `Standard`)
            lo = rand64(seed, gi);
            hi = rand64(seed ^ 0xA5A5A5A5A5A5A5A5ull, gi);
            break;
        case 1: {  // Zipf-shaped (distr.rs:54-76,108-130): continuous inverse of H(x) = (x^(1-s) - 1)/(1-s), N = 2^min(bits,64) - 1
            const uint64_t r = rand64(seed, gi);
            if (param == 1.0) {  // s = 1: x = 2^(u * bits), in fixed point (reproducible on the host)
                const uint64_t t = (r >> 32) * (uint64_t)(bits >= 64 ? 64u : bits);  // 32.32
                const uint64_t x = gen_exp2_floor((uint32_t)(t >> 32), (uint32_t)t);
                lo = x - 1;  // x >= 1
            } else {  // other exponents: double precision pow (device libm: the shape, not a host-reproducible bit pattern)
                const double u = (double)(r >> 11) * (1.0 / 9007199254740992.0);
                const double N1 = bits >= 64 ? 18446744073709551616.0 : (double)(1ull << bits);
                double x = pow(1.0 + u * (pow(N1, 1.0 - param) - 1.0), 1.0 / (1.0 - param));
                x = floor(x) - 1.0;
                if (x < 0) x = 0;
                lo = x >= 18446744073709551615.0 ? ~0ull : (uint64_t)x;
            }
            break;
        }
        case 2: {  // step-uniform over `param` equally spaced values (distr.rs:78-106,132-160)
            const uint64_t k = iparam;
            const uint64_t maxv = bits >= 64 ? ~0ull : ((1ull << bits) - 1);
            const uint64_t s = maxv / (k + 1);
            lo = s * (1 + rand64(seed, gi) % k);
            break;
        }
        case 3: lo = gi; break;
        case 4: lo = n_total - 1 - gi; break;
        case 6: {  // geometric, success probability `param` (distr.rs:3-38 MyExp): floor(log2(U) / log2(1 - p)) by inversion;
                   // iparam = -log2(1 - p) as 32.32 fixed point
            uint64_t w = rand64(seed, gi);
            if (w == 0) w = 1;
            lo = gen_neg_log2(w) / iparam;
            break;
        }
        default: lo = iparam; break;  // constant
    }
}

// payload_zero: every byte outside the key is 0 (the reference's `(key, 0)` pairs, distr.rs:22-26,42-52);
// else the payload holds the low bytes of the element's global index (reveals instability).
__global__ __launch_bounds__(256) void rsx_generate_kernel(uint8_t* __restrict__ data, uint64_t n,
                                                           uint32_t elem_bytes, uint32_t key_offset,
                                                           uint32_t key_bytes, int gen, uint64_t seed, double param,
                                                           uint64_t iparam, uint64_t index_base, uint32_t payload_zero) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t gi = index_base + i;
        uint64_t lo, hi;
        gen_key(gen, seed, param, iparam, gi, index_base + n, key_bytes, lo, hi);
        uint8_t* e = data + i * elem_bytes;
        uint32_t pb = 0;  // payload byte counter
        for (uint32_t b = 0; b < elem_bytes; ++b) {
            if (b >= key_offset && b < key_offset + key_bytes) {
                const uint32_t kb = b - key_offset;
                e[b] = (uint8_t)((kb < 8 ? lo >> (8 * kb) : hi >> (8 * (kb - 8))) & 0xFF);
            } else {
                e[b] = (pb < 8 && !payload_zero) ? (uint8_t)((gi >> (8 * pb)) & 0xFF) : 0;
                ++pb;
            }
        }
    }
}

// mapped key of element i as (hi, lo) unsigned 128-bit
__device__ __forceinline__ void mapped_key(const uint8_t* e, uint32_t key_offset, uint32_t key_bytes, uint32_t kind,
                                           uint64_t& lo, uint64_t& hi) {
    lo = hi = 0;
    const uint32_t top = key_bytes - 1;
    const bool neg = (kind == 2) && (e[key_offset + top] & 0x80);
    for (uint32_t k = 0; k < key_bytes; ++k) {
        uint32_t b = e[key_offset + k];
        if (neg) b ^= 0xFF;
        else if (k == top && kind != 0) b ^= 0x80;
        if (k < 8) lo |= (uint64_t)b << (8 * k);
        else hi |= (uint64_t)b << (8 * (k - 8));
    }
}

// Lower and upper bound of 128-bit mapped-key queries in a range that is sorted by mapped key:
// out[qi] = number of elements of the range with key < Q, out[nq + qi] = number with key <= Q.  The
// range of query qi is [ranges[2 qi], ranges[2 qi + 1]) of `data`, or all n elements when `ranges`
// is null.  One thread per query; the multi-GPU splitter search asks a few hundred at a time.
__global__ __launch_bounds__(256) void rsx_bounds_kernel(const uint8_t* __restrict__ data, uint64_t n,
                                                         uint32_t elem_bytes, uint32_t key_offset, uint32_t key_bytes,
                                                         uint32_t kind, const uint64_t* __restrict__ q, uint32_t nq,
                                                         uint64_t* __restrict__ out, const uint64_t* __restrict__ ranges) {
    const uint32_t qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= nq) return;
    const uint64_t qlo = q[2 * qi], qhi = q[2 * qi + 1];
    const uint64_t begin = ranges ? ranges[2 * qi] : 0;
    uint64_t end = ranges ? ranges[2 * qi + 1] : n;
    if (end > n) end = n;
    uint64_t lo = begin < end ? begin : end, hi = end;
    while (lo < hi) {  // first element with key >= Q
        const uint64_t mid = lo + (hi - lo) / 2;
        uint64_t klo, khi;
        mapped_key(data + mid * elem_bytes, key_offset, key_bytes, kind, klo, khi);
        if (khi < qhi || (khi == qhi && klo < qlo)) lo = mid + 1;
        else hi = mid;
    }
    out[qi] = lo - (begin < end ? begin : end);
    hi = end;
    while (lo < hi) {  // first element with key > Q
        const uint64_t mid = lo + (hi - lo) / 2;
        uint64_t klo, khi;
        mapped_key(data + mid * elem_bytes, key_offset, key_bytes, kind, klo, khi);
        if (khi < qhi || (khi == qhi && klo <= qlo)) lo = mid + 1;
        else hi = mid;
    }
    out[nq + qi] = lo - (begin < end ? begin : end);
}

// ---- splitter search, one digit per launch, no host in the loop ----------------------------------
// The multi-GPU schedules cut sorted ranges at exact global ranks: digit by digit from the top, 256 candidate keys
// per boundary are counted in every rank's range, the counts are summed over the ranks (an all-reduce) and the
// largest candidate whose global count does not exceed the boundary's rank fixes the digit.  These two kernels keep
// the key prefix on the device, so that the digits follow each other stream-ordered: count -> all-reduce -> pick.
//
// less[b * 256 + j] = elements of range b, [ranges[2b], ranges[2b+1]) of `data` (sorted by mapped key), whose key is
// below prefix[b] | j << 8*digit.  prefix[2b], prefix[2b+1] = low / high 64 bits of the mapped-key prefix.
// grid = boundaries, block = 256 (one candidate per thread).
__global__ __launch_bounds__(256) void rsx_splitter_count_kernel(const uint8_t* __restrict__ data, uint64_t n, uint32_t elem_bytes,
                                                                 uint32_t key_offset, uint32_t key_bytes, uint32_t kind,
                                                                 const uint64_t* __restrict__ ranges,
                                                                 const uint64_t* __restrict__ prefix, uint32_t digit,
                                                                 uint64_t* __restrict__ less) {
    const uint32_t b = blockIdx.x, j = threadIdx.x;
    uint64_t qlo = prefix[2 * b], qhi = prefix[2 * b + 1];
    if (digit < 8) qlo |= (uint64_t)j << (8 * digit);
    else qhi |= (uint64_t)j << (8 * (digit - 8));
    const uint64_t begin = ranges[2 * b];
    uint64_t end = ranges[2 * b + 1];
    if (end > n) end = n;
    uint64_t lo = begin < end ? begin : end, hi = end;
    const uint64_t base = lo;
    while (lo < hi) {  // first element with key >= Q
        const uint64_t mid = lo + (hi - lo) / 2;
        uint64_t klo, khi;
        mapped_key(data + mid * elem_bytes, key_offset, key_bytes, kind, klo, khi);
        if (khi < qhi || (khi == qhi && klo < qlo)) lo = mid + 1;
        else hi = mid;
    }
    less[(size_t)b * RADIX + j] = lo - base;
}

// prefix[b] |= pick << 8*digit with pick = the largest j whose summed count total[b * 256 + j] does not exceed rank[b]
// (the counts are monotone in j and total[b * 256] <= rank[b] always: it counts keys below the prefix itself)
__global__ __launch_bounds__(256) void rsx_splitter_pick_kernel(const uint64_t* __restrict__ total, const uint64_t* __restrict__ rank,
                                                                uint64_t* __restrict__ prefix, uint32_t digit) {
    __shared__ uint32_t cnt[4];
    const uint32_t b = blockIdx.x, j = threadIdx.x;
    const uint64_t ok = __ballot(total[(size_t)b * RADIX + j] <= rank[b]);
    if ((j & 63u) == 0u) cnt[j >> 6] = (uint32_t)__popcll(ok);
    __syncthreads();
    if (j == 0) {
        const uint32_t below = cnt[0] + cnt[1] + cnt[2] + cnt[3];  // candidates 0 .. below-1 qualify
        const uint64_t pick = below ? below - 1u : 0u;
        if (digit < 8) prefix[2 * b] |= pick << (8 * digit);
        else prefix[2 * b + 1] |= pick << (8 * (digit - 8));
    }
}

__global__ __launch_bounds__(256) void rsx_verify_kernel(const uint8_t* __restrict__ data, uint64_t n,
                                                         uint32_t elem_bytes, uint32_t key_offset, uint32_t key_bytes,
                                                         uint32_t kind, uint64_t* __restrict__ out) {
    uint64_t bad = 0, sum = 0, unstable = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t pay_bytes = elem_bytes - key_bytes;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint8_t* e = data + i * elem_bytes;
        uint64_t lo, hi;
        mapped_key(e, key_offset, key_bytes, kind, lo, hi);
        uint64_t h = 0x243F6A8885A308D3ull;
        for (uint32_t b = 0; b < elem_bytes; b++) h = splitmix64(h ^ e[b]);
        sum += h;
        if (i + 1 < n) {
            const uint8_t* f = e + elem_bytes;
            uint64_t lo2, hi2;
            mapped_key(f, key_offset, key_bytes, kind, lo2, hi2);
            if (hi > hi2 || (hi == hi2 && lo > lo2)) ++bad;
            if (pay_bytes && hi == hi2 && lo == lo2) {
                uint64_t p1 = 0, p2 = 0;
                uint32_t pb = 0;
                for (uint32_t b = 0; b < elem_bytes && pb < 8; ++b) {
                    if (b >= key_offset && b < key_offset + key_bytes) continue;
                    p1 |= (uint64_t)e[b] << (8 * pb);
                    p2 |= (uint64_t)f[b] << (8 * pb);
                    ++pb;
                }
                if (p1 > p2) ++unstable;
            }
        }
    }
    // wave reduce then one atomic per wave
    for (int o = 32; o > 0; o >>= 1) {
        bad += __shfl_down(bad, o);
        sum += __shfl_down(sum, o);
        unstable += __shfl_down(unstable, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if (bad) atomicAdd((unsigned long long*)&out[0], (unsigned long long)bad);
        atomicAdd((unsigned long long*)&out[1], (unsigned long long)sum);
        if (unstable) atomicAdd((unsigned long long*)&out[2], (unsigned long long)unstable);
    }
}

}  // namespace rsx
