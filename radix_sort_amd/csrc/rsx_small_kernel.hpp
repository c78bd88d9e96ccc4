// rsx_small_kernel.hpp -- arrays of at most one tile: the whole sort in ONE launch of ONE workgroup.
//
// The general path costs a memset, a count kernel and D sweeps whatever the size (>= 4.5 us each, and a
// sweep of one tile still walks its roll call, cursors and flush): 70 us for 1000 u32 keys.  Here the
// elements live in registers and LDS for all D passes of mod.rs:84-169: per pass, stable ranks inside each
// wave in (round, lane) order, a digit-major / wave-minor scan of the wave counters (count -> prefix,
// mod.rs:90-120 with chunk == wave), a scatter into LDS and a read back in tile order (mod.rs:121-168); the
// last pass is written to memory from LDS.  Same bytes as the general path (both are the stable LSD sort).
#pragma once
#include "rsx_device.hpp"

namespace rsx {

struct SmallArgs {
    void* data;            // n elements, sorted in place
    uint32_t n;            // 1 .. 512 * KPT
    uint32_t passes;       // key bytes (T::NUMBER_OF_DIGITS)
    uint32_t rank_atomic;  // ranks may come from returned LDS atomics (context self-test)
    uint32_t map_keys;     // signed / float keys: mapped on load, mapped back on store
    DigitSpec spec[16];    // digit of pass d of the MAPPED key (flip == 0)
    KeyXform xf;
};

template <int ES, int KPT>
__global__ __launch_bounds__(512) void rsx_small_sort_kernel(const SmallArgs a) {
    constexpr int WG = 512, NWAVE = WG / WAVE;
    using E = Elem<ES>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    E* s_elems = reinterpret_cast<E*>(smem);                                              // [WG * KPT]
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(smem + (size_t)WG * KPT * sizeof(E));  // [NWAVE][256]
    uint32_t* s_misc = s_cnt + NWAVE * RADIX;                                             // [NWAVE]
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t n = a.n;
    const uint32_t kp = (n + WG - 1) / WG;          // rounds in use, 1..KPT (wave-uniform, the same for all)
    const uint32_t seg = wave * (WAVE * kp) + lane;  // wave w holds elements [w*64*kp, (w+1)*64*kp), round j at +j*64:
                                                     // (wave, round, lane) order == index order, so ranks are stable
    E* __restrict__ data = static_cast<E*>(a.data);
    E e[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        e[j] = E{};
        if ((uint32_t)j < kp) {
            const uint32_t p = seg + (uint32_t)j * WAVE;
            if (p < n) {
                e[j] = data[p];
                if (a.map_keys) key_map<ES, false>(e[j], a.xf);
            }
        }
    }
    uint32_t* my = s_cnt + wave * RADIX;
    for (uint32_t pass = 0; pass < a.passes; ++pass) {
        const DigitSpec spec = a.spec[pass];
#pragma unroll
        for (int i = 0; i < RADIX / WAVE; ++i) my[i * WAVE + lane] = 0;
        // the slots past n are padding: digit 255, and being the highest indices they rank behind every real 255
        auto digit_of = [&](int j) -> uint32_t {
            return (seg + (uint32_t)j * WAVE >= n) ? 255u : elem_digit<ES, false>(e[j], spec);
        };
        uint32_t rk[KPT];
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            rk[j] = 0;
            if ((uint32_t)j < kp) {
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
        }
        __syncthreads();
        // count -> prefix: start of (digit v, wave w) in the sorted tile, digit-major / wave-minor
        uint32_t tcount = 0, incl = 0;
        if (tid < RADIX) {
#pragma unroll
            for (int w = 0; w < NWAVE; ++w) tcount += s_cnt[w * RADIX + tid];
            incl = wave_incl_scan<true>(tcount);
            if (lane == 63) s_misc[wave] = incl;
        }
        __syncthreads();
        if (tid < RADIX) {
            uint32_t run = incl - tcount;
            for (uint32_t w = 0; w < wave; ++w) run += s_misc[w];
#pragma unroll
            for (int w = 0; w < NWAVE; ++w) {
                const uint32_t c = s_cnt[w * RADIX + tid];
                s_cnt[w * RADIX + tid] = run;
                run += c;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < KPT; ++j)
            if ((uint32_t)j < kp) s_elems[my[digit_of(j)] + rk[j]] = e[j];
        __syncthreads();
        if (pass + 1 < a.passes) {
#pragma unroll
            for (int j = 0; j < KPT; ++j)
                if ((uint32_t)j < kp) e[j] = s_elems[seg + (uint32_t)j * WAVE];
            // (the next scatter into s_elems comes two barriers later)
        }
    }
    for (uint32_t i = tid; i < n; i += WG) {
        E x = s_elems[i];
        if (a.map_keys) key_map<ES, true>(x, a.xf);
        data[i] = x;
    }
}

}  // namespace rsx
