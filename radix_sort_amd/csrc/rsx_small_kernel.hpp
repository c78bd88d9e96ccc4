// rsx_small_kernel.hpp -- sorts that ONE workgroup finishes in LDS.
//
//  rsx_small_sort_kernel   arrays of at most one tile: the whole sort in ONE launch of ONE workgroup.
//  rsx_bucket_sort_kernel  the second half of a middle-size sort: the first sweep has split the array into the 256
//                          buckets of its MOST significant digit (a stable partition: bucket v sits at its final
//                          position range, in input order); workgroup v sorts bucket v by the remaining D-1
//                          digits.  Two trips through memory instead of D, and one launch instead of D-1.
//  rsx_bucket16_kernel     the last stage of the wide-key hybrid (rsx_mid_kernels.hpp): two sweeps have partitioned
//                          the array by a 16-bit window of the key; persistent workgroups sort the 65536 buckets (or
//                          groups of small ones) by the digits below it.
//  Both bucket kernels start their passes at the digit their arrays' size allows (first_digit_for, PassPlan) and mend the few
//  neighbours that still agree afterwards by the skipped digits (local_finish, mend_listed).
//
// The general path costs a memset, a count kernel and D sweeps whatever the size (>= 10 us each: a sweep of one
// tile still walks its roll call, cursors and flush): 70 us for 1000 u32 keys, 110 us for 2^20.  Here the
// elements live in registers and LDS for all passes of mod.rs:84-169: per pass, stable ranks inside each
// wave in (round, lane) order, a digit-major / wave-minor scan of the wave counters (count -> prefix,
// mod.rs:90-120 with chunk == wave), a scatter into LDS and a read back in tile order (mod.rs:121-168); the
// last pass is written to memory from LDS.  Same bytes as the general path (all are the stable LSD sort; a
// stable partition by the top digit followed by a stable LSD sort of each bucket by the lower digits is the
// stable sort by the whole key).
#pragma once
#include "rsx_device.hpp"

namespace rsx {

struct SmallArgs {
    const void* src;       // n elements (the bucket kernel: the array partitioned by its top digit)
    void* data;            // where the sorted elements go (may be src)
    uint32_t n;            // small kernel: 1 .. 512 * KPT
    uint32_t passes;       // digits to sort by: key bytes (T::NUMBER_OF_DIGITS), one less for the bucket kernel
    uint32_t rank_atomic;  // ranks may come from returned LDS atomics (context self-test)
    uint32_t map_load;     // signed / float keys: mapped on load (the bucket kernel finds them mapped)
    uint32_t map_store;    // ... and mapped back on store
    DigitSpec spec[16];    // digit of pass d of the MAPPED key (flip == 0)
    KeyXform xf;
    // bucket kernel only
    const uint32_t* top_tot;          // the 256 totals of the top digit (rsx_tilescan_kernel)
    uint32_t cap;                     // elements a 1024-thread workgroup sorts in LDS (a 256-thread one: a quarter)
    uint32_t no_skip;                 // bucket kernels: run every pass (1), or start at the digit first_digit_for() names and mend (0)
    uint32_t group_shift;             // bucket16 kernel: a workgroup takes 2^group_shift consecutive buckets as ONE array
    uint32_t key_offset, key_bytes;   // bucket kernels: where the key sits in the element (they build their compare masks: PassPlan)
    uint32_t* hint;                   // host-visible report for the host's next forecast: 1 = every bucket of this input fits a
                                      // 256-thread workgroup, 3 = a 1024-thread one, 2 = some bucket fits neither
};

// Which of the byte digits a workgroup's LDS passes run over, [first, end) (first > 0: the digits below are skipped and
// the result mended, local_finish), and the key bits from digit `first` up per element dword.  Uniform (scalar registers).
struct PassPlan {
    uint32_t first, end;
    uint32_t mask[8];  // key bytes of digits first .. (what the passes sorted by)
    uint32_t low[8];   // key bytes of digits 0 .. first-1 (what they skipped)
    __device__ __forceinline__ void set_masks(const uint32_t key_offset, const uint32_t key_bytes) {
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            uint32_t m = 0, l = 0;
#pragma unroll
            for (uint32_t b = 0; b < 4; ++b) {
                const uint32_t byte = 4u * (uint32_t)w + b;
                if (byte >= key_offset + first && byte < key_offset + key_bytes) m |= 0xFFu << (8u * b);
                if (byte >= key_offset && byte < key_offset + first) l |= 0xFFu << (8u * b);
            }
            mask[w] = m;
            low[w] = l;
        }
    }
};

// How many of an array's variable key bits its LDS passes must cover: m keys sorted by b of their bits leave about
// m^2 / 2^(b+1) pairs of neighbours that agree on them (uniform bits), and mend_listed puts right a few dozen runs in the
// time of a fraction of a pass: b = 2 ceil(log2 m) - RSX_MEND_SLACK keeps the expected pairs at 2^(RSX_MEND_SLACK - 1) or
// fewer.  The passes then start at the highest byte digit that leaves them those bits (at least one pass is run).
#ifndef RSX_MEND_SLACK
#define RSX_MEND_SLACK 6u
#endif
__device__ __forceinline__ uint32_t first_digit_for(const uint32_t m, const uint32_t variable_bits, const uint32_t end) {
    const uint32_t lg = m <= 1u ? 0u : 32u - (uint32_t)__builtin_clz(m - 1u);
    const uint32_t need = 2u * lg > RSX_MEND_SLACK ? 2u * lg - RSX_MEND_SLACK : 0u;
    uint32_t first = variable_bits > need ? (variable_bits - need) >> 3 : 0u;
    if (first + 1u > end) first = end != 0u ? end - 1u : 0u;
    return first;
}

// The per-wave digit counters of these sorts.  (16-bit halves, two to a word, for elements of 12 bytes and more would
// save 8 KiB of LDS per 1024 threads -- measured: lanes on digits 2k and 2k+1 then meet on one word, and with the extra
// shifts every size of 16-byte elements ran 8-20 % slower.  Kept as a switch; off.)
template <int ES>
struct WaveCnt {
    static constexpr bool HALF = false;
    using T = typename std::conditional<HALF, uint16_t, uint32_t>::type;
    static __device__ __forceinline__ void zero(T* my, uint32_t lane) {
        uint32_t* w = reinterpret_cast<uint32_t*>(my);
#pragma unroll
        for (int i = 0; i < (HALF ? 2 : 4); ++i) w[i * WAVE + lane] = 0;
    }
    static __device__ __forceinline__ uint32_t add_rtn(T* my, uint32_t d, uint32_t v) {  // returns the count before
        if constexpr (HALF) {
            const uint32_t sh = (d & 1u) * 16u;
            return (atomicAdd(reinterpret_cast<uint32_t*>(my) + (d >> 1), v << sh) >> sh) & 0xFFFFu;
        } else {
            return atomicAdd(&my[d], v);
        }
    }
};
// What a workgroup of WG threads x KPT registers holds in LDS: WG * KPT elements, or what 160 KiB leave beside the wave
// counters and the 3 KiB of the sort through memory (1024 threads of 16-byte elements: 9020 of 9216 slots).
template <int ES, int KPT, int WG>
__host__ __device__ constexpr uint32_t cape() {
    constexpr uint32_t slots = (uint32_t)WG * KPT;
    constexpr uint32_t room = ((163840u - 1024u - (uint32_t)(WG / WAVE) * RADIX * (uint32_t)sizeof(typename WaveCnt<ES>::T) - 64u - 3u * RADIX * 4u) / (uint32_t)ES) & ~3u;  // (1 KiB for the kernels' static LDS)
    return slots < room ? slots : room;
}

// The elements of a workgroup's array [0, n) as its threads hold them: wave w holds [w*64*kp, (w+1)*64*kp), round j at
// +j*64 -- (wave, round, lane) order == index order, so ranks are stable.  kp = ceil(n / WG) rounds are in use.
template <int ES, int KPT, int WG>
__device__ __forceinline__ void local_load(const SmallArgs& a, const Elem<ES>* __restrict__ src, const uint32_t n, Elem<ES> (&e)[KPT]) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t kp = (n + WG - 1) / WG;
    const uint32_t seg = wave * (WAVE * kp) + lane;
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        e[j] = Elem<ES>{};
        if ((uint32_t)j < kp) {
            const uint32_t p = seg + (uint32_t)j * WAVE;
            if (p < n) {
                e[j] = src[p];
                if (a.map_load) key_map<ES, false>(e[j], a.xf);
            }
        }
    }
}

// The a.passes digit passes over the elements in `e` (local_load); the sorted array is left in LDS (s_elems[0, n)).
template <int ES, int KPT, int WG>
__device__ __forceinline__ void local_passes(const SmallArgs& a, Elem<ES> (&e)[KPT], const uint32_t n, unsigned char* smem,
                                             const uint32_t first = 0, const uint32_t end = ~0u) {
    const uint32_t stop = end == ~0u ? a.passes : end;
    constexpr int NWAVE = WG / WAVE;
    using E = Elem<ES>;
    E* s_elems = reinterpret_cast<E*>(smem);                                              // [WG * KPT]
    using C = WaveCnt<ES>;
    typename C::T* s_cnt = reinterpret_cast<typename C::T*>(smem + (size_t)cape<ES, KPT, WG>() * sizeof(E));  // [NWAVE][256]
    uint32_t* s_misc = reinterpret_cast<uint32_t*>(s_cnt + NWAVE * RADIX);                         // [NWAVE]
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t kp = (n + WG - 1) / WG;          // rounds in use, 1..KPT (wave-uniform, the same for all)
    const uint32_t seg = wave * (WAVE * kp) + lane;
    typename C::T* my = s_cnt + wave * RADIX;
    for (uint32_t pass = first; pass < stop; ++pass) {
        const DigitSpec spec = a.spec[pass];
        C::zero(my, lane);
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
                    rk[j] = C::add_rtn(my, d, 1u);
                } else {
                    const uint64_t m = match_digit(d);
                    const uint32_t below = mbcnt64(m);
                    const uint32_t seen = my[d];
                    if (below == 0) (void)C::add_rtn(my, d, (uint32_t)__popcll(m));
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
                s_cnt[w * RADIX + tid] = (typename C::T)run;
                run += c;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < KPT; ++j)
            if ((uint32_t)j < kp) {
                // (padding slots, digit 255, rank last and land past n: inside LDS only if it has all WG * KPT slots)
                if constexpr (cape<ES, KPT, WG>() < (uint32_t)WG * KPT) {
                    if (seg + (uint32_t)j * WAVE < n) s_elems[(uint32_t)my[digit_of(j)] + rk[j]] = e[j];
                } else {
                    s_elems[(uint32_t)my[digit_of(j)] + rk[j]] = e[j];
                }
            }
        __syncthreads();
        if (pass + 1 < stop) {
#pragma unroll
            for (int j = 0; j < KPT; ++j)
                if ((uint32_t)j < kp) e[j] = s_elems[seg + (uint32_t)j * WAVE];
            // (the next scatter into s_elems comes two barriers later)
        }
    }
}

// s_elems[0, n) -> dst[0, n)
template <int ES, int KPT, int WG>
__device__ __forceinline__ void local_store(const SmallArgs& a, Elem<ES>* __restrict__ dst, const uint32_t n, unsigned char* smem) {
    const Elem<ES>* s_elems = reinterpret_cast<const Elem<ES>*>(smem);
    for (uint32_t i = threadIdx.x; i < n; i += WG) {
        Elem<ES> x = s_elems[i];
        if (a.map_store) key_map<ES, true>(x, a.xf);
        dst[i] = x;
    }
}

// Sorts elements [0, n) of `src` by `a.passes` digits into `dst` (same index range), n <= WG * KPT.
template <int ES, int KPT, int WG>
__device__ __forceinline__ void local_sort(const SmallArgs& a, const Elem<ES>* __restrict__ src, Elem<ES>* __restrict__ dst,
                                           const uint32_t n, unsigned char* smem) {
    Elem<ES> e[KPT];
    local_load<ES, KPT, WG>(a, src, n, e);
    local_passes<ES, KPT, WG>(a, e, n, smem);
    local_store<ES, KPT, WG>(a, dst, n, smem);
}

// A bucket LARGER than a workgroup holds (the host predicted a spread-out top digit from the previous sort's counts and
// this input is skewed): the workgroup sorts it through memory -- D-1 LSD passes (an odd number: the last lands in
// `buf1`) between the bucket's ranges of the two buffers, each pass a count, a scan and a scatter of mod.rs:90-168 with
// chunk == WG * KPT elements taken in order (stable ranks inside the chunk as in local_sort, running cursors across
// chunks).  Slow (one CU for the whole bucket) and rare: the context falls back to LSD passes for its next sorts.
// One pass through memory by ONE workgroup: src[0, m) -> dst[0, m), a stable partition by digit(x) in [0, 256) -- count,
// scan and chunk-wise stable scatter (mod.rs:90-168 with chunk == what the workgroup holds in LDS).  Thread t < 256 gets
// its digit's start and count.  map_back: signed / float keys are mapped back on the way out.
template <int ES, int KPT, int WG, typename DigitFn>
__device__ __forceinline__ void stream_pass(const SmallArgs& a, const Elem<ES>* __restrict__ src, Elem<ES>* __restrict__ dst, const uint32_t m,
                                            unsigned char* smem, DigitFn digit, const bool map_back, uint32_t& bin_start, uint32_t& bin_count) {
    constexpr int NWAVE = WG / WAVE;
    constexpr uint32_t CH = cape<ES, KPT, WG>();
    using E = Elem<ES>;
    E* s_elems = reinterpret_cast<E*>(smem);
    using C = WaveCnt<ES>;
    typename C::T* s_cnt = reinterpret_cast<typename C::T*>(smem + (size_t)cape<ES, KPT, WG>() * sizeof(E));
    uint32_t* s_misc = reinterpret_cast<uint32_t*>(s_cnt + NWAVE * RADIX);  // [NWAVE]
    uint32_t* s_gbase = s_misc + NWAVE;        // [256] where the next element of each digit goes, relative to the array
    uint32_t* s_dstart = s_gbase + RADIX;      // [256] start of each digit's run in the sorted chunk
    uint32_t* s_dcount = s_dstart + RADIX;     // [256] its length
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    typename C::T* my = s_cnt + wave * RADIX;
    if (tid < RADIX) s_gbase[tid] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < m; i += WG) atomicAdd(&s_gbase[digit(src[i])], 1u);  // count (mod.rs:90-109)
    __syncthreads();
    if (tid < RADIX) {  // prefix (mod.rs:110-120)
        const uint32_t c = s_gbase[tid];
        const uint32_t incl = wave_incl_scan<true>(c);
        if (lane == 63) s_misc[wave] = incl;
        s_dcount[tid] = incl - c;
        bin_count = c;
    }
    __syncthreads();
    if (tid < RADIX) {
        uint32_t b = s_dcount[tid];
        for (uint32_t w = 0; w < wave; ++w) b += s_misc[w];
        s_gbase[tid] = b;
        bin_start = b;
    }
    __syncthreads();
    for (uint32_t c0 = 0; c0 < m; c0 += CH) {  // scatter (mod.rs:121-168), chunk by chunk in order
        const uint32_t n = m - c0 < CH ? m - c0 : CH;
        const uint32_t kp = (n + WG - 1) / WG;
        const uint32_t seg = wave * (WAVE * kp) + lane;
        E e[KPT];
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            e[j] = E{};
            if ((uint32_t)j < kp) {
                const uint32_t p = seg + (uint32_t)j * WAVE;
                if (p < n) e[j] = src[c0 + p];
            }
        }
        C::zero(my, lane);
        auto digit_of = [&](int j) -> uint32_t { return (seg + (uint32_t)j * WAVE >= n) ? 255u : digit(e[j]); };
        uint32_t rk[KPT];
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            rk[j] = 0;
            if ((uint32_t)j < kp) {
                const uint32_t d = digit_of(j);
                const uint64_t mm = match_digit(d);  // (ballots: the rare path does not depend on the LDS ordering self-test)
                const uint32_t below = mbcnt64(mm);
                const uint32_t seen = my[d];
                if (below == 0) (void)C::add_rtn(my, d, (uint32_t)__popcll(mm));
                rk[j] = seen + below;
            }
        }
        __syncthreads();
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
            s_dstart[tid] = run;
            s_dcount[tid] = tid == RADIX - 1 ? tcount - (kp * WG - n) : tcount;  // padding slots rank as digit 255, last
#pragma unroll
            for (int w = 0; w < NWAVE; ++w) {
                const uint32_t c = s_cnt[w * RADIX + tid];
                s_cnt[w * RADIX + tid] = (typename C::T)run;
                run += c;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < KPT; ++j)
            if ((uint32_t)j < kp) {
                // (padding slots, digit 255, rank last and land past n: inside LDS only if it has all WG * KPT slots)
                if constexpr (cape<ES, KPT, WG>() < (uint32_t)WG * KPT) {
                    if (seg + (uint32_t)j * WAVE < n) s_elems[(uint32_t)my[digit_of(j)] + rk[j]] = e[j];
                } else {
                    s_elems[(uint32_t)my[digit_of(j)] + rk[j]] = e[j];
                }
            }
        __syncthreads();
        for (uint32_t i = tid; i < n; i += WG) {
            E x = s_elems[i];
            const uint32_t d = digit(x);
            const uint32_t pos = s_gbase[d] + (i - s_dstart[d]);
            if (map_back) key_map<ES, true>(x, a.xf);
            dst[pos] = x;
        }
        __syncthreads();
        if (tid < RADIX) s_gbase[tid] += s_dcount[tid];
        __syncthreads();
    }
    // whoever reads dst next does so through this CU's vector L1: write back, meet, and drop what the L1 holds
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

template <int ES, int KPT, int WG>
__device__ void big_bucket_sort(const SmallArgs& a, Elem<ES>* buf0, Elem<ES>* buf1, const uint32_t m, unsigned char* smem) {
    using E = Elem<ES>;
    E* src = buf0;
    E* dst = buf1;
    for (uint32_t pass = 0; pass < a.passes; ++pass) {
        const DigitSpec spec = a.spec[pass];
        uint32_t bs = 0, bc = 0;
        stream_pass<ES, KPT, WG>(a, src, dst, m, smem, [&](const E& x) { return elem_digit<ES, false>(x, spec); },
                                 pass + 1 == a.passes && a.map_store != 0, bs, bc);
        E* t = src;
        src = dst;
        dst = t;
    }
}

// waves per SIMD the bucket16 kernel is compiled for: its LDS lets 2 workgroups of 512 threads (4 waves per SIMD) or 3
// of 256 (3 waves per SIMD) share a CU, if their registers do
#define RSX_B16_WAVES(WG) ((WG) >= 512 ? 4 : 3)
// Wide keys (rsx_mid_kernels.hpp, rsx_count16top_kernel): the array is partitioned by a 16-bit window of the mapped key
// (WidePlan); starts[b] .. starts[b + 1] is bucket b.  Persistent workgroups take the buckets round-robin and sort each
// by the digits below the window in LDS, in place; one that does not fit goes through memory (`scratch`, same offsets).
// WG: 1024 threads (one workgroup per CU), 512 (two) or 256 (three) by the average bucket.
//
// After local_passes(first = f > 0) the array in LDS is sorted by digits f .. passes-1 only, elements that agree on
// those in input order.  Where neighbours agree (a "run"), the skipped digits 0 .. f-1 decide.
// agree(): on the key bits from digit f up (PassPlan::mask).
template <int ES>
__device__ __forceinline__ bool agree(const PassPlan& pp, const Elem<ES>& x, const Elem<ES>& y) {
    uint32_t diff = 0;
#pragma unroll
    for (int w = 0; w < ES / 4; ++w) diff |= (x.w[w] ^ y.w[w]) & pp.mask[w];
    return diff == 0;
}
// The usual case of a mend: a few runs of two or three in an array of thousands (a bucket of m keys whose passes covered
// b bits leaves m^2 / 2^(b+1) agreeing pairs).  The threads that saw a tie list the HEADS of the runs (a tie whose
// predecessor is not one) in the wave counters' LDS; one thread per listed head sorts its run by the skipped digits, and
// every thread reads its elements back.  More heads than the list holds, or a run too long for an insertion sort: every
// pass on what LDS holds (the caller).
constexpr uint32_t MEND_LIST = 192;
template <int ES, int KPT, int WG>
__device__ __forceinline__ uint32_t* mend_count(unsigned char* smem) {
    return reinterpret_cast<uint32_t*>(smem + (size_t)cape<ES, KPT, WG>() * sizeof(Elem<ES>));  // [0] heads listed, [1 ..] the list
}
// The end of a bucket whose passes started at digit f > 0: the sorted tile is read back in store order together with
// every element's predecessor (local_check; true if some neighbours agree); if none do -- the rule -- the registers go
// straight to memory (local_store_regs).
template <int ES, int KPT, int WG>
__device__ __forceinline__ bool local_check(const PassPlan& pp, const uint32_t n, unsigned char* smem, Elem<ES> (&x)[KPT], uint32_t& ties) {
    using E = Elem<ES>;
    const E* s_elems = reinterpret_cast<const E*>(smem);
    ties = 0;  // bit j: element j * WG + tid agrees with its predecessor (no short-circuit: the reads of all rounds are in flight together)
    if (threadIdx.x == 0) *mend_count<ES, KPT, WG>(smem) = 0;  // (the wave counters are free after the last pass's barrier)
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const uint32_t i = (uint32_t)j * WG + threadIdx.x;
        x[j] = E{};
        if (i < n) {
            x[j] = s_elems[i];
            const E p = s_elems[i != 0 ? i - 1 : 0];
            uint32_t diff = 0;
#pragma unroll
            for (int w = 0; w < ES / 4; ++w) diff |= (x[j].w[w] ^ p.w[w]) & pp.mask[w];
            ties |= (diff == 0 && i != 0) ? 1u << j : 0u;
        }
    }
    // Neighbours that agree on the skipped digits too are equal keys, in input order already: an array of few distinct
    // keys (every run longer than a mend would take on) goes straight to memory unless some pair really differs.
    // (Looked at again by the threads that saw a tie, in a loop of its own: one more value kept through the unrolled loop
    // above cost the 1024- and 512-thread forms 38 spilled registers.)
    uint32_t work = 0;
    for (uint32_t t = ties; t != 0; t &= t - 1u) {  // (one round per tie: a thread of a uniform array has none, or one)
        const uint32_t i = (uint32_t)__builtin_ctz(t) * WG + threadIdx.x;
        const E u = s_elems[i], v = s_elems[i - 1];
#pragma unroll
        for (int w = 0; w < ES / 4; ++w) work |= (u.w[w] ^ v.w[w]) & pp.low[w];
    }
    return __syncthreads_or(work != 0 ? 1 : 0) != 0;
}
// (Not inlined: inside the bucket kernels' loop its registers cost the 1024- and 512-thread forms 70-86 spilled ones and
// 2^30 u64 8 %.  It takes what it needs by value -- no argument block, which would go to scratch.)
#ifndef RSX_MEND_INLINE
#define RSX_MEND_INLINE __forceinline__
#endif
template <int ES>
struct MendMasks {
    uint32_t hi[ES / 4], lo[ES / 4];
};
template <int ES>
__device__ RSX_MEND_INLINE uint32_t mend_listed(Elem<ES>* s_elems, uint32_t* s_list, uint32_t* s_flag, const uint32_t n, const uint32_t wg,
                                                          uint32_t ties, const MendMasks<ES> mm) {
    using E = Elem<ES>;
    constexpr uint32_t MAX_RUN = 48;
    auto same = [&](const E& u, const E& v) {
        uint32_t diff = 0;
#pragma unroll
        for (int w = 0; w < ES / 4; ++w) diff |= (u.w[w] ^ v.w[w]) & mm.hi[w];
        return diff == 0;
    };
    auto less = [&](const E& u, const E& v) {  // by the skipped digits, the highest first: their bytes as one number
#pragma unroll
        for (int w = ES / 4 - 1; w >= 0; --w) {
            const uint32_t a = u.w[w] & mm.lo[w], b = v.w[w] & mm.lo[w];
            if (a != b) return a < b;
        }
        return false;
    };
    if (threadIdx.x == 0) *s_flag = 0;
    for (uint32_t j = 0; ties != 0; ++j, ties >>= 1) {
        if (!(ties & 1u)) continue;
        const uint32_t i = j * wg + threadIdx.x;  // agrees with i - 1; is i - 1 the head of the run?
        if (i >= 2 && same(s_elems[i - 1], s_elems[i - 2])) continue;
        const uint32_t k = atomicAdd(&s_list[0], 1u);
        if (k < MEND_LIST) s_list[1 + k] = i - 1;
    }
    __syncthreads();
    const uint32_t heads = s_list[0];
    if (heads > MEND_LIST) return 0u;  // (nothing was moved yet)
    if (threadIdx.x < heads) {
        const uint32_t i = s_list[1 + threadIdx.x];
        uint32_t len = 2;
        while (i + len < n && len <= MAX_RUN && same(s_elems[i + len], s_elems[i])) ++len;
        if (len > MAX_RUN) {
            *s_flag = 1;
        } else {
            for (uint32_t k = 1; k < len; ++k) {  // stable insertion by the skipped digits
                const E y = s_elems[i + k];
                uint32_t j = k;
                while (j > 0 && less(y, s_elems[i + j - 1])) {
                    s_elems[i + j] = s_elems[i + j - 1];
                    --j;
                }
                s_elems[i + j] = y;
            }
        }
    }
    __syncthreads();
    return *s_flag == 0 ? 1u : 0u;
}
template <int ES, int KPT, int WG>
__device__ __forceinline__ bool local_mend_listed(const uint32_t n, unsigned char* smem, const PassPlan& pp, uint32_t* s_flag, const uint32_t ties) {
    static_assert((WG / WAVE) * RADIX >= MEND_LIST + 1, "the list lives in the wave counters");
    MendMasks<ES> mm;
#pragma unroll
    for (int w = 0; w < ES / 4; ++w) {
        mm.hi[w] = pp.mask[w];
        mm.lo[w] = pp.low[w];
    }
    return mend_listed<ES>(reinterpret_cast<Elem<ES>*>(smem), mend_count<ES, KPT, WG>(smem), s_flag, n, (uint32_t)WG, ties, mm) != 0;
}
template <int ES, int KPT, int WG>
__device__ __forceinline__ void local_store_regs(const SmallArgs& a, Elem<ES>* __restrict__ dst, const uint32_t n, Elem<ES> (&x)[KPT]) {
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const uint32_t i = (uint32_t)j * WG + threadIdx.x;
        if (i < n) {
            if (a.map_store) key_map<ES, true>(x[j], a.xf);
            dst[i] = x[j];
        }
    }
}

// false: the runs could not be mended (nothing stored)
template <int ES, int KPT, int WG>
__device__ __forceinline__ bool local_finish(const SmallArgs& a, Elem<ES>* __restrict__ dst, const uint32_t n, unsigned char* smem, const PassPlan& pp,
                                             uint32_t* s_flag) {
    Elem<ES> x[KPT];
    uint32_t ties;
    if (!local_check<ES, KPT, WG>(pp, n, smem, x, ties)) {
        local_store_regs<ES, KPT, WG>(a, dst, n, x);
        return true;
    }
    if (!local_mend_listed<ES, KPT, WG>(n, smem, pp, s_flag, ties)) return false;
    local_store<ES, KPT, WG>(a, dst, n, smem);
    return true;
}

// One array of at most WG * KPT elements, src -> dst: the passes [pp.first, pp.end), the check, the mending or -- if the
// runs are too long for that -- every pass [0, pp.end) (and pp.first = 0 for the caller's later arrays).
template <int ES, int KPT, int WG>
__device__ __forceinline__ void local_sort_skip(const SmallArgs& a, const Elem<ES>* __restrict__ src, Elem<ES>* __restrict__ dst, const uint32_t n,
                                                unsigned char* smem, PassPlan& pp, uint32_t* s_flag) {
    using E = Elem<ES>;
    E e[KPT];
    local_load<ES, KPT, WG>(a, src, n, e);
    local_passes<ES, KPT, WG>(a, e, n, smem, pp.first, pp.end);
    if (pp.first == 0) {
        local_store<ES, KPT, WG>(a, dst, n, smem);
    } else if (!local_finish<ES, KPT, WG>(a, dst, n, smem, pp, s_flag)) {
        pp.first = 0;
        const uint32_t kp = (n + WG - 1) / WG, seg = (threadIdx.x >> 6) * (WAVE * kp) + (threadIdx.x & 63u);
#pragma unroll
        for (int j = 0; j < KPT; ++j)
            if ((uint32_t)j < kp) e[j] = reinterpret_cast<const E*>(smem)[seg + (uint32_t)j * WAVE];
        __syncthreads();
        local_passes<ES, KPT, WG>(a, e, n, smem, 0, pp.end);
        local_store<ES, KPT, WG>(a, dst, n, smem);
    }
}

template <int ES, int KPT>
__global__ __launch_bounds__(512) void rsx_small_sort_kernel(const SmallArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    local_sort<ES, KPT, 512>(a, static_cast<const Elem<ES>*>(a.src), static_cast<Elem<ES>*>(a.data), a.n, smem);
}

// grid = 256: workgroup v sorts the bucket of top-digit value v, [start_v, start_v + count_v) of the partitioned
// array, by the lower digits.  The bucket's place comes from the 256 totals of the top digit.
// WG: 1024 threads, or 256 for small buckets (n / 256 <= 2048 and the forecast says they all fit 256 x KPT): fewer
// waves to scan and to wait for at the barriers -- 2^16 u32 keys 15.9 -> ~10 us.
template <int ES, int KPT, int WG>
__global__ __launch_bounds__(WG) void rsx_bucket_sort_kernel(const SmallArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* s_red = reinterpret_cast<uint64_t*>(smem);  // [0..3] partial sums, [4] the bucket's count, [5..8] partial maxima
    const uint32_t tid = threadIdx.x, v = blockIdx.x;
    uint64_t c = 0;
    if (tid < RADIX) {
        c = a.top_tot[tid];
        if (tid == v) s_red[4] = c;
        if (v == 0) {  // workgroup 0 reports the largest bucket
            uint64_t big = c;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const uint64_t y = __shfl_xor(big, o);
                big = y > big ? y : big;
            }
            if ((tid & 63u) == 0u) s_red[5 + (tid >> 6)] = big;
        }
        uint64_t below = tid < v ? c : 0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) below += __shfl_xor(below, o);
        if ((tid & 63u) == 0u) s_red[tid >> 6] = below;
    }
    __syncthreads();
    const uint64_t start = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    const uint64_t count = s_red[4];
    if (v == 0 && tid == 0) {
        uint64_t big = s_red[5];
        for (int w = 1; w < 4; ++w) big = s_red[5 + w] > big ? s_red[5 + w] : big;
        __hip_atomic_store(a.hint, big <= (uint64_t)a.cap / 4 ? 1u : big <= (uint64_t)a.cap ? 3u : 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();  // smem is the sort's from here
    if (count == 0) return;
    if (count > (uint64_t)cape<ES, KPT, WG>()) {  // a skewed top digit the host did not foresee: through memory, by this workgroup alone
        big_bucket_sort<ES, KPT, WG>(a, const_cast<Elem<ES>*>(static_cast<const Elem<ES>*>(a.src)) + start,
                                     static_cast<Elem<ES>*>(a.data) + start, (uint32_t)count, smem);
        return;
    }
    if constexpr (ES < 8) {  // (keys of more than five bytes only: narrower elements never skip)
        local_sort<ES, KPT, WG>(a, static_cast<const Elem<ES>*>(a.src) + start, static_cast<Elem<ES>*>(a.data) + start, (uint32_t)count, smem);
    } else {
        // (as in rsx_bucket16_kernel: the passes start at the digit this bucket's size allows, neighbours that still agree are mended)
        PassPlan pp;
        pp.end = a.passes;
        pp.first = a.no_skip ? 0u : first_digit_for((uint32_t)count, 8u * a.passes, a.passes);
        pp.set_masks(a.key_offset, a.key_bytes);
        uint32_t* s_flag = reinterpret_cast<uint32_t*>(reinterpret_cast<typename WaveCnt<ES>::T*>(smem + (size_t)cape<ES, KPT, WG>() * sizeof(Elem<ES>)) + (WG / WAVE) * RADIX) + (WG / WAVE);
        local_sort_skip<ES, KPT, WG>(a, static_cast<const Elem<ES>*>(a.src) + start, static_cast<Elem<ES>*>(a.data) + start, (uint32_t)count, smem, pp,
                                     s_flag);
    }
}

template <int ES, int KPT, int WG>
__global__ __launch_bounds__(WG, RSX_B16_WAVES(WG)) void rsx_bucket16_kernel(const SmallArgs a, const uint64_t* __restrict__ starts, void* scratch,
                                                                         const WidePlan* __restrict__ plan, Gate gate) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (!gate_open(gate)) return;
    using E = Elem<ES>;
    constexpr int NWAVE = WG / WAVE;
    uint32_t* s_flag = reinterpret_cast<uint32_t*>(reinterpret_cast<typename WaveCnt<ES>::T*>(smem + (size_t)cape<ES, KPT, WG>() * sizeof(E)) + NWAVE * RADIX) + NWAVE;  // (s_misc is [NWAVE]; 16 words there)
    // A bucket of m elements that agree on their window (and everything above it) is, as a rule, told apart by the next
    // 2 log2(m) bits or so: the passes start at the digit that leaves them 2 log2(m) - 6 variable bits (first_digit_for:
    // three passes for the 16384-key buckets of 2^30 u64), the neighbours that still agree afterwards are put right one
    // run at a time (mend_listed), and a workgroup that meets an input where that does not work -- long runs: few distinct
    // values in those bits -- runs all passes from then on.
    // Small buckets (2^24 u64 keys: 256 each) cost a workgroup ~8 us apiece whatever they hold.  The host then sets
    // group_shift: 2^group_shift consecutive buckets -- a contiguous range of the final order -- are sorted as ONE array,
    // by the key's digits up to the window's top.  A group that does not fit after all is done bucket by bucket, every pass.
    // (Requesting the next bucket ahead of this one's check and store -- its registers are free after the last scatter --
    // would cover a 4.6 us round trip of 31 per bucket on 2^30 u64, but every form of that loop tried spilled 50-370
    // registers and ran slower.)
    const uint32_t gs = a.group_shift != 0 ? plan->group_shift : 0;  // (the host offers groups up to 2^a.group_shift; rsx_scan16_kernel chose)
    const bool medium = (plan->verdict & VERDICT_MEDIUM) != 0;
    PassPlan pp;  // (uniform: scalar loads and registers)
    pp.end = gs == 0 ? plan->pass_end : plan->group_end;
    if (pp.end == 0) {  // the window reaches the key's lowest bit: the two sweeps were the sort; signed / float keys are still mapped
        if (a.map_store) {
            const uint64_t n = starts[65536];
            E* data = static_cast<E*>(a.data);
            for (uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x; i < n; i += (uint64_t)gridDim.x * WG) {
                E x = data[i];
                key_map<ES, true>(x, a.xf);
                data[i] = x;
            }
        }
        return;
    }
    // The passes of an array (a bucket, or a group of 2^gs) run over the byte digits that hold its variable bits -- the
    // b_lo bits below the window and the gs low bits of the window -- from the digit first_digit_for() names for the
    // largest array this launch can meet (the largest bucket the scan saw, times the group, at most what LDS holds).
    const uint32_t b_lo = plan->window_top - 15u;
    const uint32_t end_sub = pp.end;              // bucket by bucket inside a group that does not fit: every digit, as before
    if (gs != 0) pp.end = (b_lo + gs + 7u) / 8u;  // (the window's higher bits are the same in the whole group)
    {
        const uint64_t big = (uint64_t)plan->scan_max << gs;
        const uint32_t m = big < (uint64_t)cape<ES, KPT, WG>() ? (uint32_t)big : cape<ES, KPT, WG>();
        pp.first = a.no_skip ? 0u : first_digit_for(m, b_lo + gs, pp.end);
    }
    pp.set_masks(a.key_offset, a.key_bytes);
    for (uint32_t g = blockIdx.x; g < (65536u >> gs); g += gridDim.x) {
        const uint32_t b0 = g << gs;
        const uint64_t gstart = starts[b0];
        const uint64_t gcount = starts[b0 + (1u << gs)] - gstart;  // (the same for every thread: uniform control flow below)
        if (gcount == 0) continue;
        const bool whole = gs == 0 || gcount <= (uint64_t)cape<ES, KPT, WG>();
        const uint32_t nsub = whole ? 1u : 1u << gs;
        for (uint32_t sub = 0; sub < nsub; ++sub) {  // (one call site each for the LDS sort and the sort through memory)
            uint64_t start = gstart, count = gcount;
            if (!whole) {
                start = starts[b0 + sub];
                count = starts[b0 + sub + 1] - start;
            }
            if (count != 0) {
                E* bucket = static_cast<E*>(a.data) + start;
                if (count <= (uint64_t)cape<ES, KPT, WG>()) {
                    const uint32_t first = pp.first, end = pp.end;
                    if (!whole) {
                        pp.first = 0;
                        pp.end = end_sub;
                    }
                    local_sort_skip<ES, KPT, WG>(a, bucket, bucket, (uint32_t)count, smem, pp, s_flag);  // (a failed mend leaves pp.first = 0)
                    if (!whole) {
                        pp.first = first;
                        pp.end = end;
                    }
                } else if (!medium) {  // (never: a bucket above what this workgroup holds makes the verdict VERDICT_MEDIUM, and
                                       // rsx_bucket16_medium_kernel takes it; without this branch the compiler spills 34 registers here)
                    big_bucket_sort<ES, KPT, WG>(a, bucket, static_cast<E*>(scratch) + start, (uint32_t)count, smem);
                }
            }
            __syncthreads();  // smem belongs to the next bucket
        }
    }
}

// The buckets ABOVE what the workgroup of the chosen form holds (a handful in an input whose density varies; all of them
// in an array too large for its 65536 buckets to fit LDS).  One workgroup per such bucket: ONE pass through memory
// (stream_pass, data -> scratch) splits it by its next k bits -- as many as bring the parts down to about 0.6 of what this
// workgroup holds, at most 8 --, then every part is sorted in LDS (scratch -> data, its final place).  A part that is
// still too large, or a bucket that eight bits do not bring down, goes through memory pass by pass (big_bucket_sort).
template <int ES, int KPT, int WG>
__global__ __launch_bounds__(WG, RSX_B16_WAVES(WG)) void rsx_bucket16_medium_kernel(const SmallArgs a, const uint64_t* __restrict__ starts, void* scratch,
                                                                                const WidePlan* __restrict__ plan, uint32_t cap256, uint32_t cap512,
                                                                                uint32_t cap1024, Gate gate) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint32_t s_part[2];
    if (!gate_open(gate)) return;
    using E = Elem<ES>;
    constexpr int NWAVE = WG / WAVE;
    constexpr uint32_t CAPE = cape<ES, KPT, WG>();
    uint32_t* s_flag = reinterpret_cast<uint32_t*>(reinterpret_cast<typename WaveCnt<ES>::T*>(smem + (size_t)CAPE * sizeof(E)) + NWAVE * RADIX) + NWAVE;
    const uint32_t tid = threadIdx.x;
    const uint32_t form = plan->verdict & VERDICT_FORM_MASK;
    const uint64_t handled = form == VERDICT_WG256 ? cap256 : form == VERDICT_WG1024 ? cap1024 : cap512;  // what the other kernel took
    const uint32_t end = plan->pass_end;
    if (end == 0) return;  // nothing below the window to sort by (the other kernel mapped the keys back)
    const uint32_t b_lo = plan->window_top - 15u;  // key bits below the window
    for (uint32_t b = blockIdx.x; b < 65536u; b += gridDim.x) {
        const uint64_t start = starts[b];
        const uint64_t count = starts[b + 1] - start;  // (the same for every thread: uniform control flow below)
        if (count <= handled) continue;
        E* bucket = static_cast<E*>(a.data) + start;
        E* scr = static_cast<E*>(scratch) + start;
        uint32_t k = 0;
        while (k < 8u && (count >> k) > (uint64_t)CAPE * 6u / 10u) ++k;
        if (k > b_lo) k = b_lo;
        // (ONE call site each for the split, the LDS sort and the sort through memory, and medium_kpt_for() elements per
        // thread: this kernel's scratch is what every hybrid sort pays for at dispatch, rsx_internal.hpp)
        const bool split = !(k == 0 || (count >> k) > (uint64_t)CAPE || count > 0xFFFFFFFFull);
        uint32_t kmask = 0, part_start = 0, part_count = (uint32_t)count;  // not split: ONE part, the bucket itself (thread 0's)
        PassPlan pp;
        pp.end = end;
        pp.first = 0;
        if (split) {
            DigitSpec sp{};
            const uint32_t ebit = 8u * a.key_offset + b_lo - k;
            sp.word = ebit >> 5;
            sp.shift = ebit & 31u;
            kmask = (1u << k) - 1u;
            part_start = part_count = 0;
            const uint32_t km = kmask;
            stream_pass<ES, KPT, WG>(a, bucket, scr, (uint32_t)count, smem, [&](const E& x) { return elem_digit_any<ES>(x, sp) & km; }, false, part_start,
                                     part_count);
            // the parts' passes: as a bucket's, one digit more when the split took half a digit of their distinguishing bits
            const uint32_t keep = plan->keep + (k >= 4u ? 1u : 0u);
            pp.first = (end > keep && !a.no_skip) ? end - keep : 0;
            pp.set_masks(a.key_offset, a.key_bytes);
        }
        for (uint32_t j = 0; j <= kmask; ++j) {
            __syncthreads();
            if (tid == j) {
                s_part[0] = part_start;
                s_part[1] = part_count;
            }
            __syncthreads();
            const uint32_t off = s_part[0], cnt = s_part[1];
            if (cnt == 0) continue;
            if (split && cnt <= CAPE) {
                local_sort_skip<ES, KPT, WG>(a, scr + off, bucket + off, cnt, smem, pp, s_flag);
            } else {
                if (split) {  // a part still too large (its keys crowd on few values of those bits): back to its place first
                    for (uint32_t i = tid; i < cnt; i += WG) bucket[off + i] = scr[off + i];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    __syncthreads();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                big_bucket_sort<ES, KPT, WG>(a, bucket + off, scr + off, cnt, smem);  // (pass by pass; an even number: ends where it began)
            }
        }
        __syncthreads();  // smem belongs to the next bucket
    }
}

}  // namespace rsx
