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

}  // namespace rsx
