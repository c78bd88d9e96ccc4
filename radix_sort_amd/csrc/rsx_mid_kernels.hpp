// rsx_mid_kernels.hpp -- the bucket split of a middle-size sort (more than one tile, up to 2^22 4-byte elements).
//
// The sweep kernel is built for arrays that keep the whole chip busy for a long time: persistent workgroups, a roll
// call, chained look-back, count matrices in replicas.  On 2^16 .. 2^22 elements those fixed costs ARE the pass
// (12 us for 2^16 keys, 25-29 us for 2^20).  The split by the most significant digit is therefore done by three
// small kernels without any hand-off between workgroups (count -> prefix -> scatter of mod.rs:90-168 with
// chunk == tile, every phase a launch of its own, as in the reference):
//   rsx_tilecount_kernel    count phase: one workgroup per tile counts the top digit of its tile: C[digit][tile]
//   rsx_tilescan_kernel     prefix phase: one workgroup per digit scans its row over the tiles (digit-major,
//                           tile-minor: mod.rs:110-120): X[tile][digit] = elements of that digit in earlier tiles;
//                           T[digit] = the digit's total
//   rsx_tilescatter_kernel  scatter phase: one workgroup per tile ranks its elements (stable, wave by wave as in
//                           local_sort), reorders them in LDS and writes every digit's run to its place
// after which rsx_bucket_sort_kernel (rsx_small_kernel.hpp) sorts each bucket by the remaining digits in LDS.
#pragma once
#include "rsx_device.hpp"

namespace rsx {

struct MidArgs {
    const void* src;
    void* dst;
    uint32_t n;
    uint32_t tiles;      // ceil(n / tile)
    uint32_t* C;         // [256][tiles] counts, digit-major
    uint32_t* X;         // [tiles][256] exclusive prefix over the earlier tiles
    uint32_t* T;         // [256] digit totals
    DigitSpec spec;      // the most significant digit: of the raw key (count kernel, with its map) / of the mapped key (scatter)
    KeyXform xf;         // scatter: signed / float keys are mapped on load and stay mapped for the bucket kernel
    uint32_t map_keys;
    uint32_t rank_atomic;
};

template <int ES, int KPT, bool FLT>
__global__ __launch_bounds__(512) void rsx_tilecount_kernel(const MidArgs a) {
    constexpr uint32_t TILE = 512u * KPT;
    __shared__ uint32_t lh[RADIX];
    const uint32_t tid = threadIdx.x, t = blockIdx.x;
    if (tid < RADIX) lh[tid] = 0;
    __syncthreads();
    const Elem<ES>* src = static_cast<const Elem<ES>*>(a.src) + (size_t)t * TILE;
    const uint32_t valid = a.n - t * TILE < TILE ? a.n - t * TILE : TILE;
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const uint32_t p = (uint32_t)j * 512u + tid;
        if (p < valid) atomicAdd(&lh[elem_digit<ES, FLT>(src[p], a.spec)], 1u);
    }
    __syncthreads();
    if (tid < RADIX) a.C[(size_t)tid * a.tiles + t] = lh[tid];
}

// grid = 256 (one digit each), block = 256 (a template only so that every element-size unit owns its copy)
template <int ES>
__global__ __launch_bounds__(256) void rsx_tilescan_kernel(const MidArgs a) {
    __shared__ uint32_t wsum[4];
    const uint32_t tid = threadIdx.x, v = blockIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t run = 0;
    for (uint32_t t0 = 0; t0 < a.tiles; t0 += 256) {
        const uint32_t t = t0 + tid;
        const uint32_t c = t < a.tiles ? a.C[(size_t)v * a.tiles + t] : 0u;
        const uint32_t incl = wave_incl_scan<true>(c);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t base = run;
        for (uint32_t w = 0; w < wave; ++w) base += wsum[w];
        if (t < a.tiles) a.X[(size_t)t * RADIX + v] = base + incl - c;
        run += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (tid == 0) a.T[v] = run;
}

// grid = tiles, block = 512; dynamic LDS: TILE * ES + 8 * 256 * 4 + 64 + 2 * 256 * 4
template <int ES, int KPT>
__global__ __launch_bounds__(512) void rsx_tilescatter_kernel(const MidArgs a) {
    constexpr int WG = 512, NWAVE = WG / WAVE;
    constexpr uint32_t TILE = (uint32_t)WG * KPT;
    using E = Elem<ES>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    E* s_elems = reinterpret_cast<E*>(smem);
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(smem + (size_t)TILE * sizeof(E));  // [NWAVE][256]
    uint32_t* s_misc = s_cnt + NWAVE * RADIX;                                          // [16]
    uint32_t* s_gbase = s_misc + 16;                                                   // [256] where this tile's run of each digit goes
    uint32_t* s_dstart = s_gbase + RADIX;                                              // [256] start of each digit's run in the sorted tile
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, t = blockIdx.x;
    const uint32_t n = a.n - t * TILE < TILE ? a.n - t * TILE : TILE;  // elements of this tile
    const E* src = static_cast<const E*>(a.src) + (size_t)t * TILE;
    E* dst = static_cast<E*>(a.dst);
    // digit starts (scan of the totals) + this tile's offset inside each digit: requested first, used last
    uint32_t tot = 0, excl = 0;
    if (tid < RADIX) {
        tot = a.T[tid];
        excl = a.X[(size_t)t * RADIX + tid];
    }
    const uint32_t seg = wave * (WAVE * KPT) + lane;  // wave-striped: (wave, round, lane) order == index order
    E e[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const uint32_t p = seg + (uint32_t)j * WAVE;
        e[j] = E{};
        if (p < n) {
            e[j] = src[p];
            if (a.map_keys) key_map<ES, false>(e[j], a.xf);
        }
    }
    uint32_t* my = s_cnt + wave * RADIX;
#pragma unroll
    for (int i = 0; i < RADIX / WAVE; ++i) my[i * WAVE + lane] = 0;
    auto digit_of = [&](int j) -> uint32_t {
        return (seg + (uint32_t)j * WAVE >= n) ? 255u : elem_digit<ES, false>(e[j], a.spec);
    };
    uint32_t rk[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const uint32_t d = digit_of(j);
        if (a.rank_atomic) {  // lanes of one instruction on one address are applied in lane order (rsx_lds_order_kernel)
            rk[j] = atomicAdd(&my[d], 1u);
        } else {
            const uint64_t m = match_digit(d);
            const uint32_t below = mbcnt64(m);
            const uint32_t seen = my[d];
            if (below == 0) atomicAdd(&my[d], (uint32_t)__popcll(m));
            rk[j] = seen + below;
        }
    }
    __syncthreads();
    uint32_t tcount = 0, incl = 0, dincl = 0;
    if (tid < RADIX) {
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) tcount += s_cnt[w * RADIX + tid];
        incl = wave_incl_scan<true>(tcount);
        dincl = wave_incl_scan<true>(tot);
        if (lane == 63) {
            s_misc[wave] = incl;
            s_misc[8 + wave] = dincl;
        }
    }
    __syncthreads();
    if (tid < RADIX) {
        uint32_t run = incl - tcount, dstart = dincl - tot;
        for (uint32_t w = 0; w < wave; ++w) {
            run += s_misc[w];
            dstart += s_misc[8 + w];
        }
        s_dstart[tid] = run;
        s_gbase[tid] = dstart + excl;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) {
            const uint32_t c = s_cnt[w * RADIX + tid];
            s_cnt[w * RADIX + tid] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KPT; ++j) s_elems[my[digit_of(j)] + rk[j]] = e[j];  // padding slots (digit 255, last) land past n
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const uint32_t i = (uint32_t)j * WG + tid;
        if (i < n) {
            const E x = s_elems[i];
            const uint32_t d = elem_digit<ES, false>(x, a.spec);
            dst[(size_t)s_gbase[d] + (i - s_dstart[d])] = x;
        }
    }
}

// ---- wide keys: two passes through memory for a 16-bit window of the key, the rest in LDS ------------------------------
// Eight-byte keys take eight sweeps of 2 n s bytes each.  A pass INSIDE LDS (rsx_small_kernel.hpp local_passes) costs a
// workgroup a fraction of what a sweep costs it per key -- no HBM, no look-back --, so when 16 bits of the mapped key
// (the window: below the highest bit in which the keys differ) spread the array over their 65536 buckets evenly enough
// for every bucket to fit a workgroup's LDS, the sort is: place the window (rsx_wideplan_kernel), count it
// (rsx_count16top_kernel + rsx_total16_kernel + rsx_scan16_kernel), two LSD sweeps on its two digits (after which every
// bucket is contiguous and in input order), and rsx_bucket16_kernel: every bucket sorted by the digits below the window
// in LDS.  Same bytes as D LSD passes (the bits above the window are the same in every key: verified by the count).
//
// rsx_wideplan_kernel: one workgroup looks at 16384 elements spread over the array (and the last one), finds the highest
// bit in which their mapped keys differ from the first element's, and places the window: the 16 bits from there down
// (its digits may lie across two dwords of the element: elem_digit_any).
// WIDEPLAN_BLOCKS workgroups of 1024 threads take one sample each (16384 in all: as ONE workgroup, 16 dependent rounds
// of loads scattered over the whole array, this kernel took 19 us at 2^23 keys and 38 us at 2^30 -- TLB misses of one CU),
// OR their differences into plan->plan_or, and the last one to finish makes the plan and clears the accumulators.
constexpr uint32_t WIDEPLAN_BLOCKS = 16;
template <int ES, bool MAP>
__global__ __launch_bounds__(1024) void rsx_wideplan_kernel(const Elem<ES>* __restrict__ src, uint64_t n, uint32_t key_offset, uint32_t key_bytes,
                                                            uint32_t key_kind, KeyXform xf, WidePlan* __restrict__ plan) {
    constexpr int NW = ES / 4;
    __shared__ uint32_t s_or[16][NW];
    __shared__ uint32_t s_last;
    const uint32_t tid = threadIdx.x;
    Elem<ES> first = src[0];
    if constexpr (MAP) key_map<ES, false>(first, xf);
    uint32_t acc[NW];
    {
        const uint64_t step = n / 16384u > 0 ? n / 16384u : 1;
        uint64_t i = ((uint64_t)blockIdx.x * 1024u + tid) * step;
        if (blockIdx.x == gridDim.x - 1 && tid == 1023) i = n - 1;
        Elem<ES> e = src[i < n ? i : 0];
        if constexpr (MAP) key_map<ES, false>(e, xf);
#pragma unroll
        for (int w = 0; w < NW; ++w) acc[w] = e.w[w] ^ first.w[w];
    }
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        uint32_t v = acc[w];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o);
        if ((tid & 63u) == 0u) s_or[tid >> 6][w] = v;
    }
    __syncthreads();
    if (tid < (uint32_t)NW) {
        uint32_t v = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) v |= s_or[q][tid];
        if (v) atomicOr(&plan->plan_or[tid], v);
    }
    __threadfence();
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(&plan->plan_done, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (s_last == 0 || tid != 0) return;
    __threadfence();
    uint32_t ors[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        ors[w] = __hip_atomic_load(&plan->plan_or[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&plan->plan_or[w], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (for the next sort)
    }
    __hip_atomic_store(&plan->plan_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // highest differing KEY bit (key bit i is element bit 8 * key_offset + i); all indices static: registers, no scratch
    const uint32_t key_lo = 8u * key_offset, key_hi = 8u * (key_offset + key_bytes) - 1u;  // element bits of the key, inclusive
    auto bits_of = [](uint32_t w, uint32_t from, uint32_t to) -> uint32_t {  // bits of dword w whose element bit index is in [from, to]
        const uint32_t lo = from > 32u * w ? from : 32u * w, hi = to < 32u * w + 31u ? to : 32u * w + 31u;
        if (lo > hi) return 0u;
        const uint32_t width = hi - lo + 1u;
        return (width == 32u ? ~0u : ((1u << width) - 1u)) << (lo - 32u * w);
    };
    int t = -1;
#pragma unroll
    for (int w = NW - 1; w >= 0; --w) {
        const uint32_t v = ors[w] & bits_of((uint32_t)w, key_lo, key_hi);
        if (t < 0 && v != 0u) t = (int)(32u * (uint32_t)w + 31u - (uint32_t)__builtin_clz(v)) - (int)key_lo;
    }
    uint32_t violation = t < 0 ? 1u : 0u;  // every sampled key the same: nothing to place a window by
    const int top = t < 15 ? 15 : t;
    const int last = (int)key_bytes * 8 - 1;
    const uint32_t lo_bit = 8u * key_offset + (uint32_t)(top - 15), hi_bit = lo_bit + 8u;
    DigitSpec lo{}, hi{};
    lo.word = lo_bit >> 5;
    lo.shift = lo_bit & 31u;
    hi.word = hi_bit >> 5;
    hi.shift = hi_bit & 31u;
    // (for the one reader of RAW keys, the count kernel of the forced mode: where the sign sits and what it flips)
    const uint32_t sign_bit = 8u * key_offset + (uint32_t)last;
    lo.top_word = hi.top_word = sign_bit >> 5;
    lo.top_shift = hi.top_shift = sign_bit & 31u;
    lo.fsign = hi.fsign = key_kind == 2u ? ~0u : 0u;
    hi.flip = (key_kind != 0u && top == last) ? 0x80u : 0u;
    plan->specs[0] = lo;
    plan->specs[1] = hi;
    plan->specs[2] = hi;
    const uint32_t b_lo = (uint32_t)(top - 15);                // key bits below the window: [0, b_lo)
    const uint32_t pass_end = (b_lo + 7u) / 8u;                // byte digits that hold them
    plan->pass_end = pass_end;
    plan->keep = (pass_end * 8u - b_lo) != 0u ? 5u : 4u;       // (rsx_bucket16_medium_kernel's parts; its top digit reaches into the window: constant bits there)
    const uint32_t group_end = ((uint32_t)top + 8u) / 8u;      // groups sort by everything up to the window's top
    plan->group_end = group_end;
    plan->group_keep = (group_end * 8u - 1u - (uint32_t)top) != 0u ? 6u : 5u;
    plan->window_top = (uint32_t)top;
    plan->violation = violation;
    for (int i = 0; i < 8; ++i) plan->scan_cnt[i * WidePlan::SCAN_LINE] = 0;
    plan->scan_done = 0;
    plan->scan_max = 0;
    plan->scan_big = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        plan->ref[w] = w < NW ? first.w[w < NW ? w : 0] : 0u;
        plan->himask[w] = top < last ? bits_of((uint32_t)w, key_lo + (uint32_t)top + 1u, key_hi) : 0u;  // key bits above the window
    }
}

// rsx_count16top_kernel: as rsx_count16_kernel (65536 16-bit LDS counters per workgroup, returned atomics, overflow
// parked in a global table), over elements of ES bytes whose bin is the plan's 16-bit window of the mapped key; every
// element is also checked against the sample above the window (plan->violation).
// Workgroup b counts chunk b % k of region b / k (k chunks per region; k == 0: the array cut into gridDim.x flat
// shares): a workgroup then stays inside ONE region of the sweeps' geometry, and the count matrix of the first sweep
// (the window's low digit, per region) is a marginal of these counters -- rsx_marginal16_kernel reads
// them back (32 MiB) instead of a count kernel reading the array again (1.37 ms of 21 on 2^30 u64).
template <int ES, bool MAP>
__global__ __launch_bounds__(1024) void rsx_count16top_kernel(const Elem<ES>* __restrict__ src, uint64_t n, WidePlan* __restrict__ plan, KeyXform xf,
                                                              uint32_t* __restrict__ P, uint32_t* __restrict__ ovf, uint32_t region_shift,
                                                              uint32_t k) {
    constexpr int NW = ES / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    uint32_t* cnt = reinterpret_cast<uint32_t*>(smem16);  // [32768]
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < 32768u / 4u; i += 1024u) reinterpret_cast<uint4*>(cnt)[i] = make_uint4(0, 0, 0, 0);
    const DigitSpec lo = plan->specs[0], hi = plan->specs[1];  // (uniform: scalar loads)
    uint32_t ref[NW], himask[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        ref[w] = plan->ref[w];
        himask[w] = plan->himask[w];
    }
    __syncthreads();
    uint64_t p0, p1;
    if (k == 0) {
        const uint64_t per = (n + gridDim.x - 1) / gridDim.x;
        p0 = (uint64_t)blockIdx.x * per;
        p1 = p0 + per;
    } else {
        const uint64_t rl = 1ull << region_shift, per = (rl + k - 1) / k;
        const uint64_t r0 = (uint64_t)(blockIdx.x / k) << region_shift;
        p0 = r0 + (uint64_t)(blockIdx.x % k) * per;
        p1 = p0 + per;
        if (p1 > r0 + rl) p1 = r0 + rl;
    }
    if (p1 > n) p1 = n;
    uint32_t stray = 0;
    auto count = [&](Elem<ES> e) {
        if constexpr (MAP) key_map<ES, false>(e, xf);
#pragma unroll
        for (int w = 0; w < NW; ++w) stray |= (e.w[w] ^ ref[w]) & himask[w];
        const uint32_t bin = (elem_digit_any<ES>(e, hi) << 8) | elem_digit_any<ES>(e, lo);
        const uint32_t sh = (bin & 1u) * 16u;
        const uint32_t old = atomicAdd(&cnt[bin >> 1], 1u << sh);
        if (((old >> sh) & 0xFFFFu) == 0x7FFFu) {  // my add made it 0x8000: park that half in the overflow table
            atomicSub(&cnt[bin >> 1], 0x8000u << sh);
            atomicAdd(&ovf[bin], 0x8000u);
        }
    };
    constexpr int UNR = ES <= 8 ? 8 : ES <= 16 ? 4 : 2;  // loads in flight per thread (one workgroup per CU: 16 waves)
    uint64_t i = p0 + tid;
    for (; i + (uint64_t)(UNR - 1) * 1024u < p1; i += (uint64_t)UNR * 1024u) {
        Elem<ES> e[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) e[u] = src[i + (uint64_t)u * 1024u];
#pragma unroll
        for (int u = 0; u < UNR; ++u) count(e[u]);
    }
    for (; i < p1; i += 1024u) count(src[i]);
    if (__syncthreads_or(stray != 0 ? 1 : 0) && tid == 0) __hip_atomic_store(&plan->violation, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint4* out = reinterpret_cast<uint4*>(P + (size_t)blockIdx.x * 32768u);
    for (uint32_t i4 = tid; i4 < 32768u / 4u; i4 += 1024u) out[i4] = reinterpret_cast<const uint4*>(cnt)[i4];
}

// The first sweep's count matrix from the 16-bit counters: J[rep][r][v] += sum over the top byte h of P[b][h * 256 + v]
// for workgroup b of rsx_count16top_kernel (region r = b / k).  Valid when no counter overflowed, which is the case
// whenever the hybrid is taken on the device's verdict: a bucket that fits a workgroup's LDS holds fewer than 0x8000
// elements.  Also does the jobs of a sort's first count kernel (count_side_jobs).  grid >= parts (the workgroups
// beyond only help with those jobs), block = 256.  (A template only so that every element-size unit owns its copy.)
template <int ES>
__global__ __launch_bounds__(256) void rsx_marginal16_kernel(const uint32_t* __restrict__ P, uint32_t parts, uint32_t k, RegionGeom g,
                                                             unsigned long long* __restrict__ J, unsigned long long* __restrict__ jclear,
                                                             uint32_t j32, uint4* __restrict__ zero16, uint64_t zero16_n, CleanList clean,
                                                             Gate gate) {
    if (!gate_open(gate)) return;
    __shared__ uint32_t part[2][RADIX];
    count_side_jobs(g, jclear, zero16, zero16_n, clean);
    if (blockIdx.x >= parts) return;
    const uint32_t tid = threadIdx.x, c = tid & 127u, half = tid >> 7;
    const uint32_t* row = P + (size_t)blockIdx.x * 32768u;
    uint32_t lo = 0, hi = 0;
#pragma unroll 8
    for (uint32_t h = half; h < 256u; h += 2u) {
        const uint32_t w = row[h * 128u + c];
        lo += w & 0xFFFFu;
        hi += w >> 16;
    }
    part[half][2u * c] = lo;
    part[half][2u * c + 1u] = hi;
    __syncthreads();
    const uint32_t sum = part[0][tid] + part[1][tid];
    const uint32_t r = blockIdx.x / k;
    const uint32_t bin = ((blockIdx.x % J_REPL) * g.num_regions + r) * RADIX + tid;
    if (sum) {
        if (j32) atomicAdd(reinterpret_cast<uint32_t*>(J) + bin, sum);
        else atomicAdd(&J[bin], (unsigned long long)sum);
    }
}

}  // namespace rsx
