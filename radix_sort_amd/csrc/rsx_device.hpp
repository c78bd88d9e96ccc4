// rsx_device.hpp -- gfx950 (CDNA4, wave64) device code for the LSD radix sort.
//
// Replaces the per-digit count -> prefix -> scatter loop of the reference
// (src/radix_sort/mod.rs:84-169) with:
//   rsx_hist_kernel     one streaming read -> all D 256-bin digit histograms
//                       (count phase, mod.rs:90-109, for every digit at once)
//   rsx_scan_kernel     exclusive scan of each 256-bin histogram (mod.rs:110-120
//                       with one chunk; the chunk-minor part is the look-back)
//   rsx_onesweep_kernel one pass: tile-local stable ranking with wave64 ballots,
//                       decoupled look-back across tiles (chunk-minor prefix),
//                       LDS reorder, coalesced run writes (mod.rs:121-168)
// Written for wave64 / 160 KiB LDS / 8 XCDs; no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rsx {

constexpr int WAVE = 64;
constexpr int RADIX = 256;

// ---------------------------------------------------------------- elements --
template <int ES>
struct Elem;  // opaque element of ES bytes, moved bitwise (mod.rs:133-140)
template <>
struct Elem<1> {
    uint8_t w[1];
};
template <>
struct Elem<2> {
    uint16_t w[1];
};
template <>
struct alignas(4) Elem<4> {
    uint32_t w[1];
};
template <>
struct alignas(8) Elem<8> {
    uint32_t w[2];
};
template <>
struct alignas(4) Elem<12> {
    uint32_t w[3];
};
template <>
struct alignas(16) Elem<16> {
    uint32_t w[4];
};
template <>
struct alignas(8) Elem<24> {
    uint32_t w[6];
};
template <>
struct alignas(16) Elem<32> {
    uint32_t w[8];
};

// Which byte of the element is the current digit, and how to map it
// (radix_digits.rs): all fields are wave-uniform kernel arguments.
struct DigitSpec {
    uint32_t word;       // dword (ES >= 4) holding the digit byte: (key_offset + digit) / 4
    uint32_t shift;      // bit offset of the digit byte inside that dword
    uint32_t top_word;   // dword holding the key's top byte (sign bit)
    uint32_t top_shift;  // bit offset of the SIGN BIT inside that dword
    uint32_t flip;       // signed/float keys: 0x80 when the digit is the top byte, else 0
};

template <int ES>
__device__ __forceinline__ uint32_t elem_word(const Elem<ES>& e, uint32_t wi) {  // wi is wave-uniform
    if constexpr (ES < 4) {
        return e.w[0];
    } else {
        constexpr int NW = ES / 4;
        uint32_t word = e.w[0];
#pragma unroll
        for (int i = 1; i < NW; ++i) word = (wi == (uint32_t)i) ? e.w[i] : word;
        return word;
    }
}

// get_digit of radix_digits.rs on one element.  FLT: f32/f64 keys (:103-124);
// otherwise unsigned (:7-53, flip == 0) or signed (:55-101, flip == 0x80 on the top byte).
template <int ES, bool FLT>
__device__ __forceinline__ uint32_t elem_digit(const Elem<ES>& e, const DigitSpec& s) {
    uint32_t d = (elem_word<ES>(e, s.word) >> s.shift) & 0xFFu;
    if constexpr (FLT) {
        const uint32_t neg = (uint32_t)((int32_t)(elem_word<ES>(e, s.top_word) << (31u - s.top_shift)) >> 31);
        d ^= (neg & 0xFFu) | (~neg & s.flip);  // negative: flip every bit; else only the sign bit
    } else {
        d ^= s.flip;
    }
    return d;
}

// generic (slow-path) byte fetch used by the histogram kernel
template <int ES>
__device__ __forceinline__ uint32_t elem_byte(const Elem<ES>& e, uint32_t b) {
    if constexpr (ES < 4) return ((uint32_t)e.w[0] >> (8 * b)) & 0xFFu;
    else return (elem_word<ES>(e, b >> 2) >> (8 * (b & 3))) & 0xFFu;
}

// ------------------------------------------------------------ wave helpers --
__device__ __forceinline__ uint32_t lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
__device__ __forceinline__ uint32_t mbcnt64(uint64_t m) {  // set bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Lanes of the wave whose 8-bit digit equals mine ("match any" on wave64):
// 8 ballots, one per digit bit.
__device__ __forceinline__ uint64_t match_digit(uint32_t d) {
    uint32_t lo = ~0u, hi = ~0u;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        uint32_t ext = (uint32_t)__builtin_amdgcn_sbfe((int)d, b, 1);  // bit b replicated to 32 bits
        asm volatile("" : "+v"(ext));  // keep the compare on `ext` (else it is re-derived from d: +1 VALU)
        const uint64_t bal = __builtin_amdgcn_ballot_w64(ext != 0);
        // m &= ~(ballot ^ mybit): lanes whose bit b equals mine; one v_bitop3 per half (LUT 0x90)
        lo = __builtin_amdgcn_bitop3_b32(lo, (uint32_t)bal, ext, 0x90);
        hi = __builtin_amdgcn_bitop3_b32(hi, (uint32_t)(bal >> 32), ext, 0x90);
    }
    return ((uint64_t)hi << 32) | lo;
}

// --------------------------------------------------------------- histogram --
// One streaming read of the input; every workgroup keeps ND 256-bin histograms
// in LDS and flushes them with one 64-bit atomic per non-empty bin.
// ghist layout: [ND][256] uint64 (digit d0 + k at row k).
template <int ES>
__global__ __launch_bounds__(512) void rsx_hist_kernel(const Elem<ES>* __restrict__ src, uint64_t n,
                                                       uint64_t* __restrict__ ghist, uint32_t key_offset,
                                                       uint32_t key_bytes, uint32_t kind, uint32_t d0,
                                                       uint32_t nd) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lh[];  // [nd][256]
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < nd * RADIX; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    const uint32_t top_byte = key_offset + key_bytes - 1;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + tid; i < n; i += stride) {
        const Elem<ES> e = src[i];
        uint32_t neg = 0;
        if (kind == 2) neg = (elem_byte<ES>(e, top_byte) & 0x80u) ? 0xFFu : 0u;
        for (uint32_t k = 0; k < nd; ++k) {
            const uint32_t b = key_offset + d0 + k;
            uint32_t d = elem_byte<ES>(e, b);
            const uint32_t flip = (b == top_byte && kind != 0) ? 0x80u : 0u;
            d ^= (kind == 2 && neg) ? 0xFFu : flip;
            atomicAdd(&lh[k * RADIX + d], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < nd * RADIX; i += blockDim.x) {
        const uint32_t c = lh[i];
        if (c) atomicAdd((unsigned long long*)&ghist[i], (unsigned long long)c);
    }
}

// In-place exclusive scan of each 256-bin row; optionally keeps the raw counts
// in `counts_out` (same layout).  One workgroup (256 threads) per row.
__global__ __launch_bounds__(256) void rsx_scan_kernel(uint64_t* __restrict__ ghist,
                                                       uint64_t* __restrict__ counts_out) {
    __shared__ uint64_t wsum[4];
    const uint32_t tid = threadIdx.x;
    uint64_t* row = ghist + (uint64_t)blockIdx.x * RADIX;
    const uint64_t c = row[tid];
    if (counts_out) counts_out[(uint64_t)blockIdx.x * RADIX + tid] = c;
    uint64_t x = c;
    const uint32_t lane = tid & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t y = __shfl_up(x, o);
        if (lane >= (uint32_t)o) x += y;
    }
    if (lane == 63) wsum[tid >> 6] = x;
    __syncthreads();
    uint64_t base = 0;
    for (uint32_t w = 0; w < (tid >> 6); ++w) base += wsum[w];
    row[tid] = base + x - c;
}

// ----------------------------------------------------------------- onesweep --
// Tile status word: [flag:2][value:BITS-2]; flag 0 = not ready, 1 = tile
// aggregate, 2 = inclusive prefix.  The word is its own flag (one relaxed
// agent-scope store/load per hop; no fence needed for a self-contained word).
template <typename S>
struct Status;
template <>
struct Status<uint32_t> {
    static constexpr int SHIFT = 30;
    static constexpr uint32_t MASK = (1u << 30) - 1;
};
template <>
struct Status<uint64_t> {
    static constexpr int SHIFT = 62;
    static constexpr uint64_t MASK = (1ull << 62) - 1;
};
template <typename S>
__device__ __forceinline__ uint64_t status_index(uint64_t tile, uint32_t digit) {
    return tile * RADIX + digit;
}

struct SweepArgs {
    const void* src;
    void* dst;
    uint64_t n;
    const uint64_t* digit_start;  // [256] exclusive global starts of this digit
    void* status;                 // [ntiles][256] status words (zeroed)
    uint32_t* ticket;             // tile ticket counter (zeroed)
    uint32_t* error;              // set non-zero if a bounded spin gave up
    DigitSpec spec;
    uint32_t dbg;                 // timing-only ablation switches (0 in production)
    unsigned long long* dbg_cnt;  // [8] diagnostic counters (dbg & 0x100)
};

// Tile = WG threads x KPT elements, held wave-striped: wave w owns the
// contiguous segment [w*64*KPT, (w+1)*64*KPT) of the tile and element j of lane
// l is segment[j*64 + l], so (wave, j, lane) order == input order and ranks
// computed in that order are stable.
template <int ES, int KPT, int WG, typename S, bool FLT>
__global__ __launch_bounds__(WG, (KPT * (ES < 4 ? 4 : ES) > 64 ? 4 : 6)) void rsx_onesweep_kernel(const SweepArgs a) {
    constexpr int NWAVE = WG / WAVE;
    constexpr int TILE = WG * KPT;
    static_assert(WG >= RADIX, "need one thread per digit");
    static_assert(TILE < (1 << 24), "rank must fit 24 bits");
    using E = Elem<ES>;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    E* s_elems = reinterpret_cast<E*>(smem);                                          // [TILE]
    uint64_t* s_base = reinterpret_cast<uint64_t*>(smem + (size_t)TILE * sizeof(E));  // [256] byte addresses
    uint32_t* s_whist = reinterpret_cast<uint32_t*>(s_base + RADIX);                  // [NWAVE][256]
    uint32_t* s_misc = s_whist + NWAVE * RADIX;                                       // [8]

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63;
    const uint32_t wave = tid >> 6;
    const E* __restrict__ src = static_cast<const E*>(a.src);
    S* status = static_cast<S*>(a.status);

    // ticket: tiles are handed out in start order, so every lower tile is already running
    if (tid == 0) s_misc[0] = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t* my_hist = s_whist + wave * RADIX;
#pragma unroll
    for (int i = 0; i < RADIX / WAVE; ++i) my_hist[i * WAVE + lane] = 0;
    __syncthreads();
    const uint64_t tile = s_misc[0];
    const uint64_t tile_base = tile * (uint64_t)TILE;
    const uint64_t remain = a.n - tile_base;
    const bool full = remain >= (uint64_t)TILE;
    const uint32_t valid = full ? (uint32_t)TILE : (uint32_t)remain;
    const uint32_t pad = TILE - valid;  // invalid tail slots, ranked as digit 255 after all valid ones

    // ---- load (wave-striped) + digit + match: all independent -> ILP ------------
    E e[KPT];
    uint32_t info[KPT];  // digit | below << 8 | count << 16, later digit | rank << 8
    const uint32_t seg = wave * (WAVE * KPT) + lane;
    if (full) {  // whole phase duplicated per branch: no pointer phis, immediates fold into the loads
        const E* p0 = src + tile_base + seg;
#pragma unroll
        for (int j = 0; j < KPT; ++j) e[j] = p0[j * WAVE];
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const uint32_t d = elem_digit<ES, FLT>(e[j], a.spec);
            const uint64_t m = (a.dbg & 8u) ? (uint64_t)d : match_digit(d);
            info[j] = d | (mbcnt64(m) << 8) | ((uint32_t)__popcll(m) << 16);
            // two matches in flight hide the SGPR-write -> VALU-read wait states; more only costs VGPRs
            if (j % 2 == 1) __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        const E* p0 = src + tile_base;
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const uint32_t p = seg + j * WAVE;
            e[j] = p0[p < valid ? p : valid - 1];  // clamped: always in bounds
        }
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            uint32_t d = elem_digit<ES, FLT>(e[j], a.spec);
            if (seg + j * WAVE >= valid) d = 255u;
            const uint64_t m = match_digit(d);
            info[j] = d | (mbcnt64(m) << 8) | ((uint32_t)__popcll(m) << 16);
            if (j % 2 == 1) __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- rank within the wave (stable): serial over rounds through the wave's LDS counters
    if (!(a.dbg & 4u))
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const uint32_t d = info[j] & 0xFFu;
        const uint32_t below = (info[j] >> 8) & 0xFFu;
        const uint32_t prev = my_hist[d];
        __builtin_amdgcn_wave_barrier();
        if (below == 0) my_hist[d] = prev + (info[j] >> 16);
        __builtin_amdgcn_wave_barrier();
        info[j] = d | ((prev + below) << 8);
    }
    __syncthreads();

    // ---- per-digit: wave counts -> tile count, publish aggregate --------------
    uint32_t cw[NWAVE];
    uint32_t tcount = 0;
    if (tid < RADIX) {
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) {
            cw[w] = s_whist[w * RADIX + tid];
            tcount += cw[w];
        }
        const uint32_t real = (tid == 255) ? tcount - pad : tcount;
        const S flag = (tile == 0) ? (S)2 : (S)1;
        __hip_atomic_store(&status[status_index<S>(tile, tid)], (flag << Status<S>::SHIFT) | (S)real, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    // exclusive scan of tcount over the 256 digits -> start of each digit's run in the tile
    uint32_t incl = tcount;
    if (tid < RADIX) {
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(incl, o);
            if (lane >= (uint32_t)o) incl += y;
        }
        if (lane == 63) s_misc[1 + wave] = incl;
    }
    __syncthreads();
    uint32_t tstart = 0;
    if (tid < RADIX) {
        uint32_t wbase = 0;
        for (uint32_t w = 0; w < wave; ++w) wbase += s_misc[1 + w];
        tstart = wbase + incl - tcount;
        uint32_t run = tstart;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) {
            s_whist[w * RADIX + tid] = run;
            run += cw[w];
        }
    }
    __syncthreads();

    // ---- reorder the tile in LDS by digit -------------------------------------
#pragma unroll
    for (int j = 0; j < KPT; ++j) s_elems[my_hist[info[j] & 0xFFu] + (info[j] >> 8)] = e[j];

    // ---- decoupled look-back: exclusive count of my digit over lower tiles ----
    if (tid < RADIX) {
        uint64_t excl = 0;
        if (tile > 0 && !(a.dbg & 1u)) {
            uint64_t p = tile - 1;
            uint32_t spins = 0, hops = 0;
            while (true) {
                ++hops;
                const S s = __hip_atomic_load(&status[status_index<S>(p, tid)], __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t f = (uint32_t)(s >> Status<S>::SHIFT);
                if (f == 0) {
                    if (++spins > (1u << 22)) {  // bounded: never hang the device
                        atomicExch(a.error, 1u);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                excl += (uint64_t)(s & Status<S>::MASK);
                if (f == 2) break;
                --p;
            }
            if ((a.dbg & 0x100u) && tid == 0) {  // diagnostics: hop / stall statistics of digit 0
                atomicAdd(&a.dbg_cnt[0], 1ull);
                atomicAdd(&a.dbg_cnt[1], (unsigned long long)hops);
                atomicAdd(&a.dbg_cnt[2], (unsigned long long)spins);
                atomicAdd(&a.dbg_cnt[3], (unsigned long long)(tile - p));
                atomicMax(&a.dbg_cnt[4], (unsigned long long)hops);
            }
            const uint32_t real = (tid == 255) ? tcount - pad : tcount;
            __hip_atomic_store(&status[status_index<S>(tile, tid)],
                               ((S)2 << Status<S>::SHIFT) | (S)((excl + real) & (uint64_t)Status<S>::MASK),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // byte address of LDS slot 0 if it belonged to this digit's run (wrap-safe in u64)
        s_base[tid] = reinterpret_cast<uint64_t>(a.dst) + (a.digit_start[tid] + excl - (uint64_t)tstart) * ES;
    }
    __syncthreads();

    // ---- write runs: consecutive threads -> consecutive addresses within a run -
    if (!(a.dbg & 2u)) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t p = i * WG + tid;
            if (full || p < valid) {
                const E x = s_elems[p];
                const uint32_t d = elem_digit<ES, FLT>(x, a.spec);
                *reinterpret_cast<E*>(s_base[d] + (uint64_t)p * ES) = x;
            }
        }
    }
}

// ------------------------------------------------------------ segmented copy --
// One workgroup walks segments grid-stride; segment copy is element-granular.
template <int ES>
__global__ __launch_bounds__(256) void rsx_segcopy_kernel(const Elem<ES>* __restrict__ src,
                                                          Elem<ES>* __restrict__ dst,
                                                          const uint64_t* __restrict__ src_off,
                                                          const uint64_t* __restrict__ dst_off,
                                                          const uint64_t* __restrict__ len, uint32_t nseg,
                                                          uint32_t blocks_per_seg) {
    const uint32_t seg = blockIdx.x / blocks_per_seg;
    const uint32_t sub = blockIdx.x % blocks_per_seg;
    if (seg >= nseg) return;
    const uint64_t L = len[seg];
    const uint64_t so = src_off[seg], dof = dst_off[seg];
    for (uint64_t i = (uint64_t)sub * blockDim.x + threadIdx.x; i < L; i += (uint64_t)blocks_per_seg * blockDim.x)
        dst[dof + i] = src[so + i];
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

// key (as up to 128 bits lo/hi) for generator `gen`
__device__ __forceinline__ void gen_key(int gen, uint64_t seed, double param, uint64_t gi, uint64_t n_total,
                                        uint32_t key_bytes, uint64_t& lo, uint64_t& hi) {
    const uint32_t bits = key_bytes * 8;
    lo = hi = 0;
    switch (gen) {
        case 0:  // uniform
            lo = rand64(seed, gi);
            hi = rand64(seed ^ 0xA5A5A5A5A5A5A5A5ull, gi);
            break;
        case 1: {  // Zipf-shaped: continuous inverse of H(x) = (x^(1-s) - 1)/(1-s), N = 2^min(bits,64) - 1
            const double u = (double)(rand64(seed, gi) >> 11) * (1.0 / 9007199254740992.0);
            const double N1 = bits >= 64 ? 18446744073709551616.0 : (double)(1ull << bits);
            double x;
            if (param == 1.0) x = exp(u * log(N1));
            else x = pow(1.0 + u * (pow(N1, 1.0 - param) - 1.0), 1.0 / (1.0 - param));
            x = floor(x) - 1.0;
            if (x < 0) x = 0;
            lo = x >= 18446744073709551615.0 ? ~0ull : (uint64_t)x;
            break;
        }
        case 2: {  // step-uniform over `param` equally spaced values (distr.rs:78-106)
            const uint64_t k = (uint64_t)param;
            const uint64_t maxv = bits >= 64 ? ~0ull : ((1ull << bits) - 1);
            const uint64_t s = maxv / (k + 1);
            lo = s * (1 + rand64(seed, gi) % k);
            break;
        }
        case 3: lo = gi; break;
        case 4: lo = n_total - 1 - gi; break;
        default: lo = (uint64_t)param; break;
    }
}

__global__ __launch_bounds__(256) void rsx_generate_kernel(uint8_t* __restrict__ data, uint64_t n,
                                                           uint32_t elem_bytes, uint32_t key_offset,
                                                           uint32_t key_bytes, int gen, uint64_t seed, double param,
                                                           uint64_t index_base) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t gi = index_base + i;
        uint64_t lo, hi;
        gen_key(gen, seed, param, gi, index_base + n, key_bytes, lo, hi);
        uint8_t* e = data + i * elem_bytes;
        uint32_t pb = 0;  // payload byte counter
        for (uint32_t b = 0; b < elem_bytes; ++b) {
            if (b >= key_offset && b < key_offset + key_bytes) {
                const uint32_t kb = b - key_offset;
                e[b] = (uint8_t)((kb < 8 ? lo >> (8 * kb) : hi >> (8 * (kb - 8))) & 0xFF);
            } else {
                e[b] = pb < 8 ? (uint8_t)((gi >> (8 * pb)) & 0xFF) : 0;
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
