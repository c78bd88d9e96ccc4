// rsx_device.hpp -- gfx950 (CDNA4, wave64) device code for the LSD radix sort.
//
// Replaces the per-digit count -> prefix -> scatter loop of the reference
// (src/radix_sort/mod.rs:84-169) with:
//   rsx_hist_kernel    count phase (mod.rs:90-109) of the first pass: digit counts per
//                      region (region == the reference's chunk)
//   rsx_sweep_kernel   prefix phase (mod.rs:110-120) in its prologue: every workgroup takes the
//                      digit-major, region-minor exclusive scan of the count matrix for the
//                      region it serves; then the
//                      scatter phase (mod.rs:121-168) of one pass: tile-local stable
//                      ranking with wave64 ballots, decoupled look-back inside each
//                      region's chain of tiles, LDS reorder, coalesced run writes;
//                      counts the NEXT pass's digit per destination region on the way
// Written for wave64 / 160 KiB LDS / 8 XCDs; no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

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
    uint32_t fsign;      // float keys: all ones (a set sign bit flips every digit), else 0
};

template <int ES>
__device__ __forceinline__ uint32_t elem_word(const Elem<ES>& e, uint32_t wi) {  // wi is wave-uniform
    if constexpr (ES < 4) {
        return e.w[0];
    } else if constexpr (ES == 4) {
        return e.w[0];
    } else if constexpr (ES == 8) {
        return wi ? e.w[1] : e.w[0];
    } else {
        // A select chain over >2 words is turned by hipcc into a dynamically indexed read of the
        // element, which puts every element into scratch memory (runtime-indexed register arrays
        // are not a thing).  A bit-select (v_bfi) on an opaque wave-uniform mask keeps them in VGPRs.
        constexpr int NW = ES / 4;
        uint32_t word = e.w[0];
#pragma unroll
        for (int i = 1; i < NW; ++i) {
            uint32_t pick = (wi == (uint32_t)i) ? ~0u : 0u;
            asm("" : "+v"(pick));  // opaque: otherwise recognised as a select again
            word = (word & ~pick) | (e.w[i] & pick);
        }
        return word;
    }
}

// get_digit of radix_digits.rs on one element.  FLT == false: the digit byte as it is --
// unsigned keys (:7-53) and every digit of a signed key except its top one.  FLT == true: the
// general map -- f32/f64 keys (:103-124: negative -> all bits flipped, else the sign bit) and the
// top digit of signed keys (:55-101: sign bit flipped); spec.fsign says which.
template <int ES, bool FLT>
__device__ __forceinline__ uint32_t elem_digit(const Elem<ES>& e, const DigitSpec& s) {
    uint32_t d = (elem_word<ES>(e, s.word) >> s.shift) & 0xFFu;
    if constexpr (FLT) {
        const uint32_t neg = (uint32_t)((int32_t)(elem_word<ES>(e, s.top_word) << (31u - s.top_shift)) >> 31) & s.fsign;
        d ^= (neg & 0xFFu) | (~neg & s.flip);
    }
    return d;
}

// A digit at ANY bit offset of the element, possibly across two dwords (the wide-key hybrid places its window below the
// highest bit in which the keys differ, wherever that is): the plain digit of a MAPPED key.  One funnel shift more than
// elem_digit; used by the hybrid's kernels only (STR instantiation of the sweep).
template <int ES>
__device__ __forceinline__ uint32_t elem_digit_any(const Elem<ES>& e, const DigitSpec& s) {
    if constexpr (ES < 8) {
        return elem_digit<ES, false>(e, s);
    } else {
        const uint32_t lo = elem_word<ES>(e, s.word);
        if (s.shift <= 24u) return (lo >> s.shift) & 0xFFu;  // (wave-uniform: the digit ends inside this dword)
        const uint32_t hi = elem_word<ES>(e, s.word + 1u);
        return (uint32_t)((((uint64_t)hi << 32) | lo) >> s.shift) & 0xFFu;
    }
}
template <int ES, bool STR>
__device__ __forceinline__ uint32_t sweep_digit(const Elem<ES>& e, const DigitSpec& s) {
    if constexpr (STR) return elem_digit_any<ES>(e, s);
    else return elem_digit<ES, false>(e, s);
}

// Signed and float keys are sorted in their order-preserving unsigned form (radix_digits.rs:
// x ^ MIN, :55-101; b ^ ((b >> 31) | MIN), :103-124).  Instead of mapping every digit of every
// pass, the FIRST pass of a sort maps the key once while the elements sit in registers and
// the LAST pass maps it back on the way out; the passes in between see plain unsigned digits.
// Per element dword i (all wave-uniform, built on the host from rsx_layout):
//   sign[i]  the key's sign bit if it lives in dword i, else 0
//   xneg[i]  bits to flip when the key counts as negative: every key bit of dword i for floats,
//            the sign bit for signed integers
//   xpos[i]  bits to flip otherwise: the sign bit
// Forward: "negative" = sign bit set.  Inverse: the mapped sign bit is the complement.
struct KeyXform {
    uint32_t sign[8];
    uint32_t xneg[8];
    uint32_t xpos[8];
};

template <int ES, bool INVERSE>
__device__ __forceinline__ void key_map(Elem<ES>& e, const KeyXform& x) {
    constexpr int NW = ES < 4 ? 1 : ES / 4;
    uint32_t sbits = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) sbits |= (uint32_t)e.w[i] & x.sign[i];
    const uint32_t neg = ((sbits != 0) != INVERSE) ? ~0u : 0u;
#pragma unroll
    for (int i = 0; i < NW; ++i)
        e.w[i] = static_cast<typename std::remove_reference<decltype(e.w[0])>::type>(
            (uint32_t)e.w[i] ^ ((neg & x.xneg[i]) | (~neg & x.xpos[i])));
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

// Lanes of the wave whose 8-bit digit equals mine ("match any" on wave64): per digit bit
// one sign-extension of the bit (v_bfe_i32), one ballot (v_cmp -> SGPR pair) and one v_bitop3
// per mask half: m &= ~(ballot ^ mybit) (LUT 0x90) -- 32 VALU per key plus the wait states
// hipcc pads between a VALU writing an SGPR and a VALU reading it.  (A hand-scheduled asm block
// with 8 distinct SGPR pairs removes the pads but costs 4-8 more live VGPRs: measured no faster,
// the kernel is latency-bound, not issue-bound.)
template <int NB = 8>
__device__ __forceinline__ uint64_t match_digit(uint32_t d) {
    uint32_t lo = ~0u, hi = ~0u;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        uint32_t ext = (uint32_t)__builtin_amdgcn_sbfe((int)d, b, 1);  // bit b replicated to 32 bits
        asm volatile("" : "+v"(ext));  // keep the compare on `ext` (else it is re-derived from d: +1 VALU)
        const uint64_t bal = __builtin_amdgcn_ballot_w64(ext != 0);
        lo = __builtin_amdgcn_bitop3_b32(lo, (uint32_t)bal, ext, 0x90);
        hi = __builtin_amdgcn_bitop3_b32(hi, (uint32_t)(bal >> 32), ext, 0x90);
    }
    return ((uint64_t)hi << 32) | lo;
}

// The same match as ONE hand-scheduled block of exactly 32 VALU: three rotating ballot
// destinations (VCC and two SGPR pairs named in the clobber list) and three rotating bit
// registers, ordered so that every v_cmp result is first read >= 3 instructions later -- the
// wait states hipcc pads with s_nop (about 10 issue slots per key in its own schedule of
// match_digit) are covered by useful work.  The sweep kernel is VALU-issue-bound in its
// match phase (removing the match altogether shortens a u32 pass by a quarter), so issue
// slots are what counts.
__device__ __forceinline__ uint64_t match_digit_sched(uint32_t d) {
    uint32_t lo, hi, t0, t1, t2;
    asm volatile(
        "v_bfe_i32 %2, %5, 0, 1\n\t"
        "v_bfe_i32 %3, %5, 1, 1\n\t"
        "v_cmp_ne_u32_e64 vcc, 0, %2\n\t"
        "v_cmp_ne_u32_e64 s[76:77], 0, %3\n\t"
        "v_bfe_i32 %4, %5, 2, 1\n\t"
        "v_cmp_ne_u32_e64 s[78:79], 0, %4\n\t"
        "v_bitop3_b32 %0, -1, vcc_lo, %2 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, -1, vcc_hi, %2 bitop3:0x90\n\t"
        "v_bfe_i32 %2, %5, 3, 1\n\t"
        "v_cmp_ne_u32_e64 vcc, 0, %2\n\t"
        "v_bitop3_b32 %0, %0, s76, %3 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, %1, s77, %3 bitop3:0x90\n\t"
        "v_bfe_i32 %3, %5, 4, 1\n\t"
        "v_cmp_ne_u32_e64 s[76:77], 0, %3\n\t"
        "v_bitop3_b32 %0, %0, s78, %4 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, %1, s79, %4 bitop3:0x90\n\t"
        "v_bfe_i32 %4, %5, 5, 1\n\t"
        "v_cmp_ne_u32_e64 s[78:79], 0, %4\n\t"
        "v_bitop3_b32 %0, %0, vcc_lo, %2 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, %1, vcc_hi, %2 bitop3:0x90\n\t"
        "v_bfe_i32 %2, %5, 6, 1\n\t"
        "v_cmp_ne_u32_e64 vcc, 0, %2\n\t"
        "v_bitop3_b32 %0, %0, s76, %3 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, %1, s77, %3 bitop3:0x90\n\t"
        "v_bfe_i32 %3, %5, 7, 1\n\t"
        "v_cmp_ne_u32_e64 s[76:77], 0, %3\n\t"
        "v_bitop3_b32 %0, %0, s78, %4 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, %1, s79, %4 bitop3:0x90\n\t"
        "v_bitop3_b32 %0, %0, vcc_lo, %2 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, %1, vcc_hi, %2 bitop3:0x90\n\t"
        "v_bitop3_b32 %0, %0, s76, %3 bitop3:0x90\n\t"
        "v_bitop3_b32 %1, %1, s77, %3 bitop3:0x90"
        : "=&v"(lo), "=&v"(hi), "=&v"(t0), "=&v"(t1), "=&v"(t2)
        : "v"(d)
        : "vcc", "s76", "s77", "s78", "s79");
    return ((uint64_t)hi << 32) | lo;
}

// ----------------------------------------------------------------- regions --
// Every pass sees its INPUT as up to MAX_REGIONS equal position ranges ("regions"):
//   region r = elements [r << region_shift, min((r + 1) << region_shift, n)).
// Regions play the "chunk" of the reference (mod.rs:66-70): a count matrix
// J[r][v] = number of elements of region r whose digit is v is known BEFORE the pass
// (count phase, mod.rs:90-109), its digit-major / region-minor exclusive scan gives
// every region its write cursors (prefix phase, mod.rs:110-120), and inside a region
// the tiles chain by decoupled look-back.  Splitting the look-back into independent
// chains matters on this chip: an agent-scope status read is a memory-side round trip
// (the 8 XCD L2s are not coherent), ~0.5 us under load, and a single chain has ~40
// tiles in the aggregate-only state at any time -- more status traffic than key traffic.
constexpr int MAX_REGIONS = 32;
// A count matrix is kept in J_REPL replicas, [J_REPL][num_regions][256]: a workgroup adds its counts to replica
// blockIdx % J_REPL (its XCD, as blocks are dealt), readers sum the replicas.  Device-scope atomics on one
// 64-byte line serialise at the memory side (~12 ns each): with one matrix, 512 workgroups flushing 2048
// counters each queued 512 deep on every line at the end of a pass (measured: 13 us per workgroup).
constexpr int J_REPL = 8;
static_assert(J_REPL == 8, "readers unroll the replica rows by 8");
// Per-pass control words: [0, MAX_REGIONS) per-region tile tickets (dynamic mode), then the roll call:
// ROLL_DONE counts complete shards, ROLL_MODE is the verdict, and ROLL_SHARD_COUNT arrival counters each
// on a line of its own (workgroup b arrives at shard b % 8: 64 arrivals per line instead of 512 on one).
constexpr int ROLL_DONE = MAX_REGIONS;
constexpr int ROLL_MODE = MAX_REGIONS + 1;
constexpr int ROLL_SHARDS = 64;            // first shard word (the words above fill one 256-byte block)
constexpr int ROLL_SHARD_COUNT = 8;
constexpr int ROLL_SHARD_STRIDE = 32;      // words: 128 bytes per shard

struct RegionGeom {
    uint64_t n;
    uint32_t region_shift;  // log2(elements per region): a multiple of the tile size
    uint32_t num_regions;   // ceil(n >> region_shift), 1..MAX_REGIONS
    uint32_t tile;          // elements per tile of the sweeps that use this geometry (host side only)
};

// what the count kernel zeroes on its way, beside its own sort's status words: the parts of the PREVIOUS sort's control
// block that it used (control words, top-digit count matrix, count matrix 0): rsx.hip begin_control
struct CleanList {
    uint4* p[3];
    uint64_t n16[3];
};
// Large arrays of wide keys are enqueued as TWO alternative kernel sequences (the 16-bit bucket hybrid and the LSD passes);
// which one runs is decided on the device by rsx_scan16_kernel: a gated kernel returns at once unless *word == value
// ((*word & mask) == value; word == null: no gate).  A launch that returns at once costs ~5 us, nothing beside these sorts' milliseconds.
struct Gate {
    const uint32_t* word;
    uint32_t mask, value;
};
__device__ __forceinline__ bool gate_open(const Gate& g) {
    if (g.word == nullptr) return true;
    return ((uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(g.word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & g.mask) == g.value;
}
// The verdict word of the wide-key hybrid (WidePlan::verdict): bit 0 the hybrid runs, bit 1 the LSD passes; with bit 0,
// bits 4-5 say which form of the bucket kernel (all are enqueued, each behind its own gate): the smallest workgroup that
// holds all but a handful of the buckets.
constexpr uint32_t VERDICT_HYBRID = 1u, VERDICT_LSD = 2u, VERDICT_PATH_MASK = 3u;
constexpr uint32_t VERDICT_MEDIUM = 0x40u;  // some buckets exceed the chosen form's workgroup: rsx_bucket16_medium_kernel takes those
constexpr uint32_t VERDICT_WG256 = 0x00u, VERDICT_WG512 = 0x10u, VERDICT_WG1024 = 0x20u, VERDICT_GROUPS = 0x30u, VERDICT_FORM_MASK = 0x30u;

// The plan of a wide-key hybrid sort, made on the device from a sample of the array (rsx_wideplan_kernel) and read by
// every kernel of the hybrid: WHICH 16 bits of the mapped key the array is partitioned by -- the two 8-bit digits below
// the highest bit in which the sampled keys differ (keys of a narrow range, value ranges of a multi-GPU sort: the bits
// above are the same for all, which rsx_count16top_kernel verifies on every element) -- and which byte digits are left
// for the LDS passes.
struct WidePlan {
    uint32_t verdict;     // rsx_scan16_kernel: VERDICT_* (the Gate word of both sequences)
    uint32_t violation;   // some element differs from the sample above the window (or the sample cannot place one): refuse
    uint32_t pass_end;    // single buckets: the LDS passes are byte digits [pass_end - keep or 0, pass_end) of the key
    uint32_t keep;
    uint32_t group_end;   // groups of buckets: digits [group_end - group_keep or 0, group_end)
    uint32_t group_keep;
    uint32_t window_top;  // key bit index of the window's top bit (diagnostic)
    uint32_t group_shift; // rsx_scan16_kernel: groups of 2^group_shift buckets (the largest the host offers whose groups fit)
    DigitSpec specs[3];   // [0] low, [1] high digit of the window (plain digits of the MAPPED key); [2] filler (sweeps read pairs)
    uint32_t ref[8];      // the mapped first element
    uint32_t himask[8];   // key bits above the window, per element dword
    // rsx_scan16_kernel's workgroups add up here, every word on a 128-byte line of its own (atomics on ONE line are
    // serialised by the memory side at ~12 ns apiece: 256 workgroups x 11 words on one line were most of that kernel's 26 us):
    // scan_cnt[i * SCAN_LINE]: [0..2] buckets above what 256 / 512 / 1024 threads hold, [3..7] groups of 2^g buckets, g = 2 .. 6,
    // above what 512 hold; scan_done: how many workgroups have added theirs (all zeroed by the plan kernel)
    static constexpr uint32_t SCAN_LINE = 32;
    uint32_t scan_cnt[8 * SCAN_LINE];
    uint32_t scan_done;
    uint32_t pad_done[SCAN_LINE - 1];
    uint32_t scan_max;    // the largest bucket (saturated)
    uint32_t pad_max[SCAN_LINE - 1];
    unsigned long long scan_big;  // elements in buckets above what the largest workgroup holds
    uint32_t pad_big[SCAN_LINE - 2];
    uint32_t plan_or[8];  // rsx_wideplan_kernel's workgroups OR their samples' differences here; the last one reads and clears
    uint32_t plan_done;   // ... how many have (zero between sorts: set once when the buffer is made, cleared by the last workgroup)
};

// --------------------------------------------------------------- histogram --
#ifndef RSX_HIST_NT
// The count kernel reads its input with the streaming hint: it arrives behind a sort whose output still sits,
// dirty, in the caches, and plain loads allocate there (evicting dirty lines while reading): 256M u32
// 0.218 -> 0.190 ms (5.66 TB/s), 1B u64 1.48 -> 1.36 ms.  The first sweep then finds a little less of its input
// cached (+0.5 %); net -0.9 % per 256M-key sort.  (The same hint on the sweep's loads costs 1-4 %, on its
// stores 80 %: partial lines are then not merged.)
#define RSX_HIST_NT 1
#endif
#ifndef RSX_HIST_UNROLL
#define RSX_HIST_UNROLL 1  // more loads in flight per thread measured slower (0.217 -> 0.228 ms per 2^28 u32 at 4)
#endif
// What a sort's first count kernel does on its way (one launch instead of four):
__device__ __forceinline__ void count_side_jobs(const RegionGeom& g, unsigned long long* __restrict__ jclear, uint4* __restrict__ zero16,
                                                uint64_t zero16_n, const CleanList& clean) {
    const uint32_t tid = threadIdx.x;
    // the count matrix the first sweep accumulates into (the second pass's) is cleared here
    if (jclear != nullptr)
        for (uint32_t i = blockIdx.x * blockDim.x + tid; i < (uint32_t)J_REPL * g.num_regions * RADIX; i += gridDim.x * blockDim.x)
            jclear[i] = 0;
    // ... and so are the tile status words of the first sweep (a memset launch less per sort)
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + tid; i < zero16_n; i += (uint64_t)gridDim.x * blockDim.x)
        zero16[i] = make_uint4(0, 0, 0, 0);
    // ... and the control block the PREVIOUS sort on this context used (the next one will find it clean: no memset launch)
#pragma unroll
    for (int z = 0; z < 3; ++z)
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + tid; i < clean.n16[z]; i += (uint64_t)gridDim.x * blockDim.x)
            clean.p[z][i] = make_uint4(0, 0, 0, 0);
}

// Count phase for ONE digit with chunk == region: J[r][v] for the pass's input.
// grid = num_regions * blocks_per_region; a block stays inside one region.
// (Only the first pass of a sort needs this kernel: each sweep pass counts the
// next pass's digit per destination region while it scatters.)
// TWO: a second digit (spec2) is counted into a second matrix (J2) on the same read -- the middle-size path
// counts the least significant digit (its first LSD pass) and the most significant one (its bucket split).
template <int ES, bool FLT, bool TWO = false>
__global__ __launch_bounds__(512) void rsx_hist_kernel(const Elem<ES>* __restrict__ src, RegionGeom g,
                                                       DigitSpec spec, uint32_t blocks_per_region,
                                                       unsigned long long* __restrict__ J,
                                                       unsigned long long* __restrict__ jclear, uint32_t j32,
                                                       uint4* __restrict__ zero16, uint64_t zero16_n,
                                                       CleanList clean, Gate gate,
                                                       DigitSpec spec2 = DigitSpec{}, unsigned long long* __restrict__ J2 = nullptr,
                                                       const DigitSpec* __restrict__ spec_dev = nullptr) {
    if (!gate_open(gate)) return;
    if (spec_dev != nullptr) spec = *spec_dev;  // the hybrid's forced mode: the digit comes from the device's plan
    // One-byte elements make 16 LDS atomics per 16-byte load, and random bins collide on the 32 banks (72 % of this
    // kernel's LDS cycles were bank conflicts: profiles/r03_u8-256m_pmc.txt): their histogram is kept in 32 copies,
    // copy l in bank l -- lane l of either half-wave adds to lh[bin * 32 + l % 32], so an instruction never has two
    // lanes on one bank or one address, whatever the skew, and needs no duplicate check.
    constexpr bool BANKED = ES == 1 && !TWO;
    __shared__ uint32_t lh[BANKED ? 32 * RADIX : TWO ? 2 * RADIX : RADIX];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < (BANKED ? 32u * RADIX : TWO ? 2u * RADIX : (uint32_t)RADIX); i += blockDim.x) lh[i] = 0;
    count_side_jobs(g, jclear, zero16, zero16_n, clean);
    __syncthreads();
    const uint32_t r = blockIdx.x / blocks_per_region;
    const uint32_t sub = blockIdx.x % blocks_per_region;
    const uint64_t begin = (uint64_t)r << g.region_shift;
    uint64_t end = begin + (1ull << g.region_shift);
    if (end > g.n) end = g.n;
    auto count = [&](const Elem<ES>& e) {
        uint32_t d;
        if (spec_dev != nullptr) {  // (the hybrid's forced mode: a window digit at any bit offset, of the RAW key)
            d = elem_digit_any<ES>(e, spec);
            if constexpr (FLT) {
                const uint32_t neg = (uint32_t)((int32_t)(elem_word<ES>(e, spec.top_word) << (31u - spec.top_shift)) >> 31) & spec.fsign;
                d ^= (neg & 0xFFu) | (~neg & spec.flip);
            }
        } else {
            d = elem_digit<ES, FLT>(e, spec);
        }
        if constexpr (BANKED) {
            atomicAdd(&lh[d * 32u + (tid & 31u)], 1u);
            return;
        }
        // skewed inputs put whole waves on one bin: count the wave with one atomic then
        const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
        const uint64_t same = __ballot(d == d0);
        if (same == __builtin_amdgcn_read_exec()) {
            if (mbcnt64(same) == 0) atomicAdd(&lh[d0], (uint32_t)__popcll(same));
        } else {
            atomicAdd(&lh[d], 1u);
        }
        if constexpr (TWO) atomicAdd(&lh[RADIX + elem_digit<ES, FLT>(e, spec2)], 1u);
    };
    // order inside a region does not matter for a count: 16-byte loads
    constexpr int VEC = (ES == 1 || ES == 2 || ES == 4 || ES == 8) ? 16 / ES : 1;
    struct alignas(VEC > 1 ? 16 : alignof(Elem<ES>)) Pack {
        Elem<ES> e[VEC];
    };
    // The C-ABI only asks for element alignment (and the multi-GPU drivers pass interior pointers): the elements
    // ahead of the first 16-byte boundary are counted one by one, like the tail.
    uint64_t head = 0;
    if constexpr (VEC > 1) {
        head = ((16u - (uint32_t)(reinterpret_cast<uintptr_t>(src + begin) & 15u)) & 15u) / ES;
        if (head > end - begin) head = end - begin;
    }
    const uint64_t nvec = (end - begin - head) / VEC;
    const Pack* vsrc = reinterpret_cast<const Pack*>(src + begin + head);
    const uint64_t stride = (uint64_t)blocks_per_region * blockDim.x;
    constexpr int UNR = RSX_HIST_UNROLL;
    for (uint64_t i = (uint64_t)sub * blockDim.x + tid; i < nvec; i += stride * UNR) {
        Pack p[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (i + u * stride < nvec) {
                if constexpr (RSX_HIST_NT != 0 && sizeof(Pack) == 16 && alignof(Pack) >= 16) {  // streaming 16-byte loads
                    typedef uint32_t v4u __attribute__((ext_vector_type(4)));  // (12-, 24-byte elements in pieces: +1 %, left plain)
                    const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(vsrc + i + u * stride));
                    __builtin_memcpy(&p[u], &v, 16);
                } else {
                    p[u] = vsrc[i + u * stride];
                }
            }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (i + u * stride < nvec) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) count(p[u].e[k]);
            }
    }
    if (VEC > 1 && sub == 0) {  // head and tail of the region (fewer than VEC elements each): wave 0 / wave 1
        if (tid < head) count(src[begin + tid]);
        const uint64_t t0 = begin + head + nvec * VEC;
        if (tid >= 64 && t0 + (tid - 64) < end) count(src[t0 + (tid - 64)]);
    }
    __syncthreads();
    if (tid < RADIX) {
        uint32_t c = 0;
        if constexpr (BANKED) {
#pragma unroll
            for (uint32_t k = 0; k < 32; ++k) c += lh[tid * 32u + ((k + tid) & 31u)];  // thread t starts at bank t: no conflicts
        } else {
            c = lh[tid];
        }
        const uint32_t bin = ((blockIdx.x % J_REPL) * g.num_regions + r) * RADIX + tid;
        if (c) {  // counters are 32 bit where a region holds < 2^32 elements (half the atomic bytes): status32()
            if (j32) atomicAdd(reinterpret_cast<uint32_t*>(J) + bin, c);
            else atomicAdd(&J[bin], (unsigned long long)c);
        }
        if constexpr (TWO) {
            const uint32_t c2 = lh[RADIX + tid];
            if (c2) {
                if (j32) atomicAdd(reinterpret_cast<uint32_t*>(J2) + bin, c2);
                else atomicAdd(&J2[bin], (unsigned long long)c2);
            }
        }
    }
}

// -------------------------------------------------------------------- sweep --
// Tile status word: [flag:2][value:BITS-2]; flag 0 = not ready, 1 = tile
// aggregate, 2 = inclusive prefix (relative to the tile's region).  The word is its
// own flag (one relaxed agent-scope store/load per hop; a self-contained word needs
// no fence).
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

struct SweepArgs {
    const void* src;
    void* dst;
    RegionGeom g;
    const unsigned long long* J;  // [J_REPL][num_regions][256] (entries of type S) this pass's count matrix (count phase, mod.rs:90-109): every
                                  // workgroup derives its region's write cursors from it (prefix phase, mod.rs:110-120)
    void* status;                 // [num_regions << (region_shift - log2 TILE)][256], zeroed
    uint32_t* tickets;            // this pass's control words, zeroed: per-region tile tickets and the roll call
                                  // (ROLL_DONE, ROLL_MODE, ROLL_SHARDS above)
    const uint32_t* prev_mode;    // the previous pass's verdict word of shard 0 (null on a first pass): a roll call that
                                  // failed there is not waited for again here (each workgroup reads its own shard's copy)
    uint16_t wg_first[MAX_REGIONS + 1];  // static mode: region r is served by workgroups [wg_first[r], wg_first[r+1])
    void* status_clean;           // the next pass's status words: every tile zeroes its row there (null on a last pass)
    uint32_t rank_atomic;         // 1: ranks may come from returned LDS atomics (ordering self-test passed)
    uint32_t hot_lanes;           // a digit shared by this many lanes of round 0 sends the tile down the ballot path
    uint32_t tiles_per_region;    // ceil(region length / tile): status rows per region
    uint32_t local_mask;          // static mode: regions whose workgroups all sit in one residue class of blockIdx % 8
    unsigned long long* jnext;    // [J_REPL][num_regions][256] next pass's count matrix (accumulated), or null
    unsigned long long* jzero;    // same shape: count matrix of the pass after next, cleared here, or null
    uint32_t* error;              // host-visible (pinned) word: set non-zero if a bounded spin gave up
    DigitSpec spec;               // this pass's digit
    DigitSpec next;               // next pass's digit (when jnext != null)
    const DigitSpec* spec_dev;    // non-null: both digits are read from device memory instead (WidePlan::specs)
    KeyXform xf;                  // signed/float key map applied on load (XF & 1) / undone on store (XF & 2)
    uint32_t opts;                // alternative paths, all bit-exact: SWEEP_OPT_*
    // middle-size path (MID instantiation: the sort's first sweep).  The host has decided, from what the previous
    // middle-size sort of the context reported, whether this launch is the bucket split by the most significant digit
    // (mid_mode 1: rsx_bucket_sort_kernel then finishes the sort) or the first LSD pass (mid_mode 2).  Either way it
    // reports through `mid_hint` (host-visible) whether every bucket of THIS input would fit a workgroup of the bucket
    // kernel (1) or not (2): the host's forecast for the next sort.
    const unsigned long long* mid_J;  // count matrix of the most significant digit
    DigitSpec mid_spec;           // the most significant digit (of the mapped key)
    uint32_t mid_cap;             // largest bucket rsx_bucket_sort_kernel sorts in LDS
    uint32_t mid_mode;
    uint32_t* mid_hint;
    Gate gate;                    // wide keys: this sweep belongs to one of two alternative sequences
    uint32_t dbg;                 // RSX_TUNING builds: timing-only ablation switches (0 in production)
    unsigned long long* dbg_cnt;  // [8] diagnostic counters (RSX_TUNING, dbg & 0x100)
};
constexpr uint32_t SWEEP_OPT_DYNAMIC = 1u;       // ticketed tiles, no roll call
constexpr uint32_t SWEEP_OPT_NO_XCD_MAJOR = 2u;  // workgroups numbered by plain blockIdx
constexpr uint32_t SWEEP_OPT_AGENT_STATUS = 4u;  // agent-scope status stores on every chain
constexpr uint32_t SWEEP_OPT_RANK_CHECK = 8u;    // cross-check one round of atomic ranks per tile against the ballots
constexpr uint32_t SWEEP_OPT_PREREAD = 16u;      // 4-byte and narrower elements: read part of the write-out from LDS ahead of the look-back
#ifdef RSX_TUNING
#define RSX_DBG(a, bit) ((a).dbg & (bit))
#else
#define RSX_DBG(a, bit) 0u
#endif

// Persistent workgroups pull tiles region by region (rotating, so consecutive tiles of
// one chain start far apart in time).  Tile = WG threads x KPT elements, held
// wave-striped: wave w owns the contiguous segment [w*64*KPT, (w+1)*64*KPT) of the
// tile and element j of lane l is segment[j*64 + l], so (wave, j, lane) order == input
// order and ranks computed in that order are stable.
//
// Per tile: load -> digit + wave64 match -> rank (wave counters in LDS) -> per-digit tile
// counts (publish aggregate) -> scan -> LDS reorder -> look-back in the region's chain ->
// run writes.  The ticket of the NEXT tile is drawn mid-tile and its loads are issued
// before the write-out, so HBM latency hides behind the look-back and the stores.
#ifdef RSX_STAMPS  // diagnostic build only: per-phase cycle shares of wave 0 (never in production)
#define RSX_STAMP(k)                                                   \
    do {                                                               \
        const unsigned long long _t = __builtin_amdgcn_s_memtime();    \
        stamp_acc[k] += _t - stamp_prev;                               \
        stamp_prev = _t;                                               \
    } while (0)
#else
#define RSX_STAMP(k) do {} while (0)
#endif
#ifndef RSX_MINW
#define RSX_MINW 6
#endif
#ifndef RSX_WO_PREREAD
#define RSX_WO_PREREAD 16  // elements of the write-out read from LDS ahead of the look-back (4-byte and narrower elements)
#endif
#ifndef RSX_WO_PREREAD8
#define RSX_WO_PREREAD8 0  // the same for 8-byte elements
#endif
#ifndef RSX_LB_WINDOW
#define RSX_LB_WINDOW 1  // look-back words requested per round trip: 2 / 4 / 8 measured +0.4 / +1.9 / +5.7 % per 256M-key sort
#endif
#ifndef RSX_START_STAGGER
#define RSX_START_STAGGER 224
#endif
#ifndef RSX_NUM_SGPR
#define RSX_NUM_SGPR 102
#endif
#ifndef RSX_RANK_GROUP
#define RSX_RANK_GROUP 4
#endif
#ifndef RSX_MINW_BIG
#define RSX_MINW_BIG 4  // waves/SIMD asked of the compiler for 64-byte-per-thread tiles (u32 x 16)
#endif
#ifndef RSX_WIDE_CNT
#define RSX_WIDE_CNT 1
#endif
#ifndef RSX_WO_GROUP
#define RSX_WO_GROUP 4  // write-out: elements in flight between scheduling barriers
#endif
#ifndef RSX_EARLY_HOP
#define RSX_EARLY_HOP 2  // 0 off, 1 on, 2 = where measured faster (elements of <= 4 bytes)
#endif
#ifndef RSX_MINW_1024
#define RSX_MINW_1024 8
#endif
#ifndef RSX_LDS_SWIZZLE
#define RSX_LDS_SWIZZLE 1
#endif
#ifndef RSX_DPP_SCAN
#define RSX_DPP_SCAN 2  // 0 off, 1 on, 2 = where measured faster (elements of <= 4 bytes)
#endif
#ifndef RSX_ROLLCALL_TICKS
#define RSX_ROLLCALL_TICKS 40000  // s_memtime ticks (shader cycles) a workgroup waits for the full grid
#endif
#ifndef RSX_LB_OVERLAP
#define RSX_LB_OVERLAP 0  // 1: walk the look-back chain beside the LDS reorder (measured slower, see below)
#endif
#ifndef RSX_LB_EVERY
#define RSX_LB_EVERY 4    // elements a wave reorders between two look-back steps
#endif
#ifndef RSX_HOT_MBCNT
#define RSX_HOT_MBCNT 1  // dominant-digit ranks: the running count rides in v_mbcnt's addend
#endif
#ifndef RSX_PREFETCH_ALL
#define RSX_PREFETCH_ALL 3  // 0 off, 1 before/after the look-back, 2 behind it, 3 = 2 where measured faster (>= 12-byte elements)
#endif

// Makes the compiler forget what it knows about the element registers: digits derived from them
// are then re-derived where needed (2 VALU) instead of being kept live across phases (1 VGPR each).
template <int ES, int KPT>
__device__ __forceinline__ void forget(Elem<ES> (&e)[KPT]) {
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        if constexpr (ES < 4) {
            uint32_t t = e[j].w[0];
            asm volatile("" : "+v"(t));
            e[j].w[0] = (decltype(e[j].w[0]))t;
        } else {
#pragma unroll
            for (int k = 0; k < ES / 4; ++k) asm volatile("" : "+v"(e[j].w[k]));
        }
    }
}

// One count into the next pass's count matrix (LDS).  Uniform keys: one LDS atomic per lane.
// Skewed keys put many lanes of a wave on one bin, and same-address LDS atomics serialise; when
// at least a quarter of the wave shares the first lane's bin the wave matches its bins instead
// (12 bits: region | digit) and one lane per bin adds the group size.
// Inclusive scan over the 64 lanes of a wave in 6 DPP steps (no LDS traffic, unlike __shfl_up):
// row_shr 1,2,4,8 scan each row of 16; row_bcast:15 adds a row's total to the next odd row,
// row_bcast:31 adds the lower half's total to the upper half.
template <bool DPP>
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
  if constexpr (DPP) {
    auto step = [](uint32_t v, auto ctrl, auto rows) {
        return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, decltype(ctrl)::value, decltype(rows)::value, 0xf, false);
    };
    x = step(x, std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});
    x = step(x, std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});
    x = step(x, std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});
    x = step(x, std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});
    x = step(x, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});
    x = step(x, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});
    return x;
  } else {
    const uint32_t lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o);
        if (lane >= (uint32_t)o) x += y;
    }
    return x;
  }
}

__device__ __forceinline__ void count_next(uint32_t* s_jn, uint32_t bin) {
    // careful form (crowded bins): the lanes that share the first lane's bin add their number once,
    // the others add 1 each -- same-address LDS atomics of one instruction are serialised
    const uint32_t b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)bin);
    const uint64_t same = __ballot(bin == b0);
    if (bin == b0) {
        if (mbcnt64(same) == 0) atomicAdd(&s_jn[b0], (uint32_t)__popcll(same));
    } else {
        atomicAdd(&s_jn[bin], 1u);
    }
}

template <int ES, int KPT>
__device__ __forceinline__ void load_tile(Elem<ES> (&e)[KPT], const Elem<ES>* __restrict__ tile_ptr, uint32_t seg,
                                          uint32_t valid, bool full) {
    if (full) {  // immediates fold into the loads
        const Elem<ES>* p0 = tile_ptr + seg;
#pragma unroll
        for (int j = 0; j < KPT; ++j) e[j] = p0[j * WAVE];
    } else {
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            // guarded, not clamped: a clamped address would be if-converted into the full path and
            // cost 2 VGPRs per load there; the partial tile is the last one of a region, so rare
            const uint32_t p = seg + j * WAVE;
            if (p < valid) e[j] = tile_ptr[p];
            else e[j] = Elem<ES>{};
        }
    }
}

template <int ES, int KPT, int WG, typename S, int XF, bool NEXT, bool MID = false, bool STR = false>
// VGPR budget: 4-byte (and narrower) keys carry 16 elements per thread and need ~104 VGPRs; capping
// them at 80 (3 workgroups/CU) spills, and the spills cost more than the third workgroup buys
// (measured: 0.82 -> 0.68 ms per 256M-key pass at 2 workgroups/CU without spills).
// SGPR budget: the hardware admits waves by SGPRs too (800 per SIMD in blocks of 16, +16 per wave):
// above 80 SGPRs a kernel cannot have 8 waves per SIMD however few VGPRs it uses.
__global__ __launch_bounds__(WG, (WG == 1024 ? RSX_MINW_1024 : ES == 16 ? 6 : KPT * (ES < 4 ? 4 : ES) >= 64 ? RSX_MINW_BIG : RSX_MINW))
__attribute__((amdgpu_num_sgpr(RSX_NUM_SGPR))) void rsx_sweep_kernel(const SweepArgs a) {
    constexpr int NWAVE = WG / WAVE;
    constexpr int TILE = WG * KPT;
    // next-tile prefetch keeps the element registers live through the write-out: only where that fits
    // the VGPR budget without spilling (u32 x 16 keys does not at 80 VGPRs, and is VALU-bound anyway)
    constexpr bool EARLY_HOP = RSX_EARLY_HOP == 1 || (RSX_EARLY_HOP == 2 && ES <= 4);
    constexpr bool PREFETCH = RSX_PREFETCH_ALL != 0 && (RSX_PREFETCH_ALL != 3 || ES >= 12);
    static_assert(WG >= RADIX, "need one thread per digit");
    static_assert(TILE <= 65536 / 2, "wave counters are 16 bit");
    using E = Elem<ES>;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr size_t TILE_BYTES = (size_t)TILE * sizeof(E);
    E* s_elems = reinterpret_cast<E*>(smem);                                           // [TILE]
    // per-wave digit counters: 32-bit where LDS allows (narrow elements run 2 workgroups/CU), else
    // two 16-bit counters per word (saves 4 KiB, costs ~6 VALU per element for shifts and masks)
    constexpr bool WIDE_CNT = RSX_WIDE_CNT && ES <= 4 && KPT >= 16 && WG <= 512;  // where the workgroup runs 2 per CU anyway
    using Cnt = typename std::conditional<WIDE_CNT, uint32_t, uint16_t>::type;
    uint32_t* s_whist2 = reinterpret_cast<uint32_t*>(smem + TILE_BYTES);               // the counters as LDS words
    Cnt* s_whist = reinterpret_cast<Cnt*>(s_whist2);                                   // [NWAVE][256]
    uint64_t* s_base = reinterpret_cast<uint64_t*>(s_whist2);                          // [256], aliases s_whist (dead by then)
    uint32_t* s_jn = s_whist2 + NWAVE * RADIX * sizeof(Cnt) / 4;                       // [num_regions][256] (NEXT)
    uint32_t* s_misc = s_jn + (NEXT ? a.g.num_regions * RADIX : 0);                    // [32]
    static_assert(NWAVE * RADIX * sizeof(Cnt) >= RADIX * sizeof(uint64_t), "s_base must fit in s_whist");

    if (!gate_open(a.gate)) return;
    const E* __restrict__ src = static_cast<const E*>(a.src);
    S* status = static_cast<S*>(a.status);
    const uint32_t NR = a.g.num_regions;
    const uint32_t tpr = a.tiles_per_region;  // status rows per region (the last tile of a region may be partial)
    const uint64_t region_len = 1ull << a.g.region_shift;

    if (NEXT)
        for (uint32_t i = threadIdx.x; i < a.g.num_regions * RADIX; i += WG) s_jn[i] = 0;
    // the count matrix of the pass after next is cleared here: its last readers (the previous pass)
    // are done, its next writers (the next pass) have not started
    using JT = S;  // count-matrix entries: 32 bit with 32-bit status words (a region then holds <= 2^30 elements)
    if (a.jzero != nullptr)
        for (uint32_t i = blockIdx.x * WG + threadIdx.x; i < (uint32_t)J_REPL * NR * RADIX; i += gridDim.x * WG)
            reinterpret_cast<JT*>(a.jzero)[i] = 0;

    // workgroup index, XCD-major (blocks are dealt round-robin over the 8 XCDs): the workgroups of
    // one chain then share an XCD, which makes their status hand-offs faster -- never a
    // correctness matter
    const uint32_t bx = (gridDim.x % 8u == 0u && !(a.opts & SWEEP_OPT_NO_XCD_MAJOR)) ? (blockIdx.x % 8u) * (gridDim.x / 8u) + blockIdx.x / 8u : blockIdx.x;
    uint32_t home = 0;
    while (home + 1 < NR && bx >= a.wg_first[home + 1]) ++home;
    const uint32_t st_step = a.wg_first[home + 1] - a.wg_first[home];  // M: workgroups serving my region
    uint32_t st_k = bx - a.wg_first[home];                             // my next tile in static mode
    uint32_t st_nt;
    {
        const uint64_t rbeg = (uint64_t)home << a.g.region_shift;
        const uint64_t rlen = (a.g.n - rbeg) < region_len ? (a.g.n - rbeg) : region_len;
        st_nt = (uint32_t)((rlen + TILE - 1) / TILE);
    }
    // (The first tile's loads are NOT issued ahead of the roll call: memory operations return in order,
    // so the roll call's polls would queue behind 28 HBM loads per lane and time out -- measured: the
    // grid then fell back to ticketed tiles, 0.46 -> 0.58 ms per 256M-key pass.)
    E e[KPT];
    bool preloaded = false;  // static mode: this tile's loads were issued during the previous tile
    // ---- who sorts which tile ----------------------------------------------------------
    // Tiles of a region form a chain and must START in order (a tile only waits for lower tiles
    // of its chain).  Two ways to guarantee that:
    //  * dynamic: an agent-scope ticket counter per region (always safe; costs an atomic round
    //    trip of 1.5-3.5 us at every tile start, ~19 % of a tile);
    //  * static: workgroup j of the M serving a region takes tiles j, j+M, j+2M, ... -- no
    //    atomics and the next tile is known early, but it needs every workgroup of the grid to
    //    be resident at once.  That is established, not assumed: a roll call at kernel start.
    //    Every workgroup adds itself to a counter and polls it for a bounded time; the first
    //    to see the full count (or to time out) fixes the mode for everybody with one CAS.
    //    A full count proves all workgroups are running, and a running workgroup stays
    //    resident until it exits.
    //    The bound: s_memtime counts shader cycles on gfx950 (~2.1-2.4 GHz), a fully resident grid
    //    answers within a few microseconds (a 1024-block grid starts first to last within 0.7 us),
    //    so 40000 ticks (~18 us, ~4 % of a 256M-key pass) is ample; a grid that is not co-resident
    //    (two streams, a co-tenant) costs that once per sort: the verdict carries over to the
    //    later passes through prev_mode.
    // ---- prefix phase (mod.rs:110-120), first half: the counts ----------------------------------
    // Write cursor of (region r, digit v) at the region's start: the digit-major, region-minor
    // exclusive running sum of the count matrix,
    //     sum_{v' < v} sum_r' J[r'][v']  +  sum_{r' < r} J[r'][v].
    // Every workgroup derives the 256 cursors of the region it serves itself (J_REPL x <= 32 x 256
    // counts); thread v keeps digit v's cursor in registers for all its tiles.  The loads are spread
    // over all 512 threads (thread t: digit t & 255, replicas of half t >> 8), up to 32 in flight each, and
    // for the home region they are issued AHEAD of the roll call, so they land while thread 0 polls.
    uint64_t rbase = 0;
    uint64_t* s_scan = reinterpret_cast<uint64_t*>(s_misc + 16);  // [4]
    uint64_t* s_half = reinterpret_cast<uint64_t*>(s_elems);       // [2][256]: the tile area is free between tiles
    static_assert(TILE_BYTES >= 2 * RADIX * sizeof(uint64_t), "scratch of the prefix phase lives in the tile area");
    auto load_counts = [&](const unsigned long long* Jm, uint32_t r, uint64_t& tot, uint64_t& below) {  // partial column sums of thread t
        tot = 0;
        below = 0;
        constexpr uint32_t HALVES = WG >= 2 * RADIX ? 2 : 1;
        constexpr uint32_t RPH = J_REPL / HALVES;  // replicas per thread
        uint32_t t = threadIdx.x;
        asm volatile("" : "+v"(t));  // opaque: the column address is not worth a register pair across the tile loop
        if (t < HALVES * RADIX) {
            const JT* col = reinterpret_cast<const JT*>(Jm) + (size_t)(t >> 8) * RPH * NR * RADIX + (t & 255u);
            constexpr uint32_t QU = ES >= 16 ? 4 : 8;  // loads in flight: RPH * QU (the 16-byte kernels run at 80 VGPRs)
            for (uint32_t q0 = 0; q0 < NR; q0 += QU) {
#pragma unroll
                for (uint32_t qq = 0; qq < QU; ++qq) {
                    const uint32_t q = q0 + qq;
                    if (q < NR) {
#pragma unroll
                        for (uint32_t rep = 0; rep < RPH; ++rep) {
                            const uint64_t c = col[(size_t)(rep * NR + q) * RADIX];
                            tot += c;
                            below += q < r ? c : 0ull;
                        }
                    }
                }
            }
        }
    };
    auto scan_cursors = [&](uint64_t tot, uint64_t below) {  // every thread calls (barriers inside)
        const uint32_t t = threadIdx.x;
        if (WG >= 2 * RADIX && t >= RADIX && t < 2 * RADIX) {
            s_half[t - RADIX] = tot;
            s_half[t] = below;
        }
        __syncthreads();
        if (WG >= 2 * RADIX && t < RADIX) {
            tot += s_half[t];
            below += s_half[RADIX + t];
        }
        uint64_t x = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t y = __shfl_up(x, o);
            if ((t & 63u) >= (uint32_t)o) x += y;
        }
        if (t < RADIX && (t & 63u) == 63u) s_scan[t >> 6] = x;
        __syncthreads();
        if (t < RADIX) {
            uint64_t run = x - tot + below;
            for (uint32_t w = 0; w < (t >> 6); ++w) run += s_scan[w];
            rbase = run;
        }
        __syncthreads();
    };
    // MID: which digit this launch partitions by, and from which count matrix, is decided below
    DigitSpec spec = a.spec;
    DigitSpec next = a.next;
    if (a.spec_dev != nullptr) {  // (uniform: scalar loads)
        spec = a.spec_dev[0];
        next = a.spec_dev[1];
    }
    const unsigned long long* Jsel = a.J;
    bool msd = false;
    auto region_cursors = [&](uint32_t r) {  // r is wave-uniform
        uint64_t tot, below;
        load_counts(Jsel, r, tot, below);
        scan_cursors(tot, below);
    };
    uint64_t tot_home, below_home;
    if constexpr (MID) {
        // Middle-size sort, first sweep: the bucket split by the MOST significant digit (a.mid_mode == 1:
        // rsx_bucket_sort_kernel then sorts every bucket by the remaining digits -- two trips through memory instead
        // of D) or the ordinary first LSD pass (2).  The largest bucket of the top digit is reported to the host.
        load_counts(a.mid_J, home, tot_home, below_home);
        const uint32_t t = threadIdx.x;
        if (WG >= 2 * RADIX && t >= RADIX && t < 2 * RADIX) s_half[t - RADIX] = tot_home;
        __syncthreads();
        uint64_t c = 0;
        if (t < RADIX) c = tot_home + (WG >= 2 * RADIX ? s_half[t] : 0ull);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint64_t y = __shfl_xor(c, o);
            c = y > c ? y : c;
        }
        if (t < RADIX && (t & 63u) == 0u) s_scan[t >> 6] = c;
        __syncthreads();
        uint64_t big = s_scan[0];
#pragma unroll
        for (int w = 1; w < RADIX / WAVE; ++w) big = s_scan[w] > big ? s_scan[w] : big;
        const uint32_t fits = (uint32_t)__builtin_amdgcn_readfirstlane((int)(big <= (uint64_t)a.mid_cap / 4 ? 1u : big <= (uint64_t)a.mid_cap ? 3u : 2u));
        msd = a.mid_mode == 1u;
        __syncthreads();  // s_half / s_scan are used again by scan_cursors
        if (msd) {
            spec = a.mid_spec;
            Jsel = a.mid_J;
        } else {
            load_counts(a.J, home, tot_home, below_home);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(a.mid_hint, fits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
        load_counts(a.J, home, tot_home, below_home);
    }
    const bool do_next = NEXT && !(MID && msd);  // the bucket split has no next pass to count for
#ifdef RSX_STAMPS
    const unsigned long long stamp_entry = __builtin_amdgcn_s_memtime();
#endif
    if (threadIdx.x == 0) {
        uint32_t mode = 2;
        const uint32_t shard = blockIdx.x % (uint32_t)ROLL_SHARD_COUNT;
        uint32_t* my_shard = a.tickets + ROLL_SHARDS + shard * ROLL_SHARD_STRIDE;  // [0] arrivals, [1] the verdict, once known
        if (!(a.opts & SWEEP_OPT_DYNAMIC) &&
            !(a.prev_mode != nullptr &&
              __hip_atomic_load(a.prev_mode + shard * ROLL_SHARD_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2u)) {
            uint32_t* done = a.tickets + ROLL_DONE;
            uint32_t* modew = a.tickets + ROLL_MODE;
            // Arrivals are counted per shard (blockIdx % 8, each on its own line); the workgroup that completes a
            // shard reports it to `done`; the one that completes `done` decides and writes the verdict into every
            // shard's line, where that shard's workgroups poll for it.  The same words also collect whether workgroups
            // sit where the XCD-major numbering assumes (XCC id == blockIdx % 8): low half = arrivals (shards
            // done), high half = workgroups (shards) that do not.
            uint32_t xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            const bool placed = (gridDim.x % 8u == 0u) && ((xcc & 0xFu) == blockIdx.x % 8u);
            const uint32_t shard_size = (gridDim.x + (uint32_t)ROLL_SHARD_COUNT - 1u - shard) / (uint32_t)ROLL_SHARD_COUNT;
            const uint32_t shards = gridDim.x < (uint32_t)ROLL_SHARD_COUNT ? gridDim.x : (uint32_t)ROLL_SHARD_COUNT;
            const uint32_t mine = placed ? 1u : 0x10001u;
            const uint32_t before = __hip_atomic_fetch_add(my_shard, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            uint32_t verdict = 2u;  // what this workgroup proposes: 2 = timed out
            bool decide = false;
            if (((before + mine) & 0xFFFFu) == shard_size) {  // my shard is complete: report it
                const uint32_t rep = ((before + mine) >> 16) ? 0x10001u : 1u;
                const uint32_t seen = __hip_atomic_fetch_add(done, rep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + rep;
                if ((seen & 0xFFFFu) == shards) {  // ... and it was the last one: everybody is running
                    verdict = seen == shards ? 3u : 1u;
                    decide = true;
                }
            }
            if (!decide) {  // wait for the verdict to appear in my shard's line (64 pollers per line, not 512 on one)
                do {
                    mode = __hip_atomic_load(my_shard + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (mode != 0u) break;
                    __builtin_amdgcn_s_sleep(4);
                } while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)RSX_ROLLCALL_TICKS);
                decide = mode == 0u;  // timed out: propose dynamic mode
            }
            if (decide) {  // the mode word is the one source of truth: first proposal wins, then it is broadcast to the shards
                uint32_t expected = 0;  // (a failed exchange leaves the winner's verdict here)
                const bool won = __hip_atomic_compare_exchange_strong(modew, &expected, verdict, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                                      __HIP_MEMORY_SCOPE_AGENT);
                mode = won ? verdict : expected;
                for (uint32_t k = 0; k < shards; ++k)
                    __hip_atomic_store(a.tickets + ROLL_SHARDS + k * ROLL_SHARD_STRIDE + 1, mode, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if (!(a.opts & SWEEP_OPT_DYNAMIC)) {
            // carried-over failure: record it for the pass after this one as well
            __hip_atomic_store(my_shard + 1, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_misc[3] = mode;
    }
    __syncthreads();
    const uint32_t mode = __builtin_amdgcn_readfirstlane(s_misc[3]);  // 1/3 static, 2 dynamic, 3 = placement verified
    const bool static_mode = mode != 2u;
    if (RSX_DBG(a, 0x100u) && threadIdx.x == 0 && blockIdx.x == 0) {
        atomicAdd(&a.dbg_cnt[5], static_mode ? 1ull : 0ull);
        atomicAdd(&a.dbg_cnt[6], mode == 3u ? 1ull : 0ull);
    }
#ifdef RSX_STAMPS
    const unsigned long long stamp_called = __builtin_amdgcn_s_memtime();
#endif
    uint32_t cur_reg = home;
    scan_cursors(tot_home, below_home);
    // Start-up stagger.  The workgroups of a chain settle into an even spread of phases over one tile period
    // (tile t+1 trails tile t by a status hand-off), which is what lets the loads of some overlap the ranking
    // and the stores of others.  Leaving the roll call they are all in the SAME phase -- every CU loading, then
    // every CU storing -- until the look-back has pulled them apart.  Workgroup j of a chain therefore starts
    // j * RSX_START_STAGGER cycles late (about half the steady-state spacing): 256M u32 2.066 -> 2.041 ms,
    // 16M u32 222 -> 209 us; 100..300 cycles measured alike, 400 worse.
    if (static_mode && RSX_START_STAGGER > 0) {
        const unsigned long long ts = __builtin_amdgcn_s_memtime();
        const unsigned long long wait = (unsigned long long)st_k * RSX_START_STAGGER;
        while (__builtin_amdgcn_s_memtime() - ts < wait) __builtin_amdgcn_s_sleep(4);
    }
    // A chain whose workgroups were all verified on ONE XCD shares one L2: its status words can then
    // be plain stores that stay in that L2 (an agent-scope store writes through to memory and the
    // next agent-scope load of the line misses: 750 vs 510 cycles per hand-off, tools/microbench/pingpong.hip).
    // The loads stay agent-scope (they bypass the reader's L1 and hit the L2).  Speed only: without
    // the proof every status store is agent-scope.
    const bool local_chain = mode == 3u && ((a.local_mask >> home) & 1u) != 0u && !(a.opts & SWEEP_OPT_AGENT_STATUS);
    auto publish = [&](S* p, S v) {
        if (local_chain) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    uint32_t rr = home;  // dynamic mode: region the next ticket is drawn from
    uint32_t alive = NR;            // regions not yet seen exhausted
    uint32_t exhausted = 0;         // bitmask of exhausted regions
    // resolve ticket `k` of region `r` into a tile, drawing further tickets while regions run dry
    auto resolve = [&](uint32_t r, uint32_t k, uint32_t slot) {
        bool have = false;
        while (alive > 0) {
            const uint64_t rbeg = (uint64_t)r << a.g.region_shift;
            const uint64_t rlen = (a.g.n - rbeg) < region_len ? (a.g.n - rbeg) : region_len;
            const uint32_t nt = (uint32_t)((rlen + TILE - 1) / TILE);
            if (k < nt) {
                have = true;
                break;
            }
            if (!(exhausted >> r & 1u)) {
                exhausted |= 1u << r;
                --alive;
            }
            if (alive == 0) break;
            do r = (r + 1 == NR) ? 0 : r + 1;
            while (exhausted >> r & 1u);
            k = __hip_atomic_fetch_add(&a.tickets[r], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_misc[slot + 0] = have ? 1u : 0u;
        s_misc[slot + 1] = r;
        s_misc[slot + 2] = k;
        if (have) rr = r;  // stay on this chain until it runs dry (rotating chains bunch up: measured slower)
    };
#ifdef RSX_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long lbs_hops = 0, lbs_spins = 0, lbs_cycles = 0, lbs_tiles = 0;  // look-back of digit 0 (thread 0)
    unsigned long long stamp_prev = __builtin_amdgcn_s_memtime();
    const unsigned long long stamp_loop = stamp_prev;
#endif
    while (true) {
        // thread coordinates are re-derived per tile from an opaque copy of threadIdx: otherwise
        // every tid-derived address (dozens of VGPRs) is hoisted out of this loop and spilled
        uint32_t tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const uint32_t lane = tid & 63;
        const uint32_t wave = tid >> 6;
        uint32_t* my_hist2 = s_whist2 + wave * (RADIX * sizeof(Cnt) / 4);
        Cnt* my_hist = s_whist + wave * RADIX;
#pragma unroll
        for (int i = 0; i < (int)(RADIX * sizeof(Cnt) / 8 / WAVE); ++i)  // each wave clears its 256 counters
            reinterpret_cast<uint64_t*>(my_hist2)[i * WAVE + lane] = 0;
        // dynamic mode: the ticket is drawn when the tile starts, so ticket order == start order inside
        // a chain.  (Drawing it earlier hides the atomic's round trip but makes a workgroup sit on a
        // ticket while the tiles behind it in the chain already wait for its aggregate: measured slower.)
        uint32_t reg, kt;
        if (static_mode) {  // wave-uniform
            __syncthreads();
            RSX_STAMP(0);
            if (st_k >= st_nt) break;
            reg = home;
            kt = st_k;
            st_k += st_step;
        } else {
            if (tid == 0)
                resolve(rr, __hip_atomic_fetch_add(&a.tickets[rr], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), 0);
            __syncthreads();
            RSX_STAMP(0);
            if (__builtin_amdgcn_readfirstlane(s_misc[0]) == 0) break;
            reg = __builtin_amdgcn_readfirstlane(s_misc[1]);
            kt = __builtin_amdgcn_readfirstlane(s_misc[2]);
            if (reg != cur_reg) {  // wave-uniform: this workgroup moved on to another region's chain
                region_cursors(reg);
                cur_reg = reg;
            }
        }

        const uint64_t tile_base = ((uint64_t)reg << a.g.region_shift) + (uint64_t)kt * TILE;
        const uint64_t rend = ((uint64_t)(reg + 1) << a.g.region_shift) < a.g.n ? ((uint64_t)(reg + 1) << a.g.region_shift) : a.g.n;
        const uint64_t remain = rend - tile_base;
        const bool full = remain >= (uint64_t)TILE;
        const uint32_t valid = full ? (uint32_t)TILE : (uint32_t)remain;
        const uint32_t pad = TILE - valid;  // invalid tail slots, ranked as digit 255 after all valid ones
        const uint64_t stat_row = ((uint64_t)reg * tpr + kt) * RADIX;
        const uint32_t seg = wave * (WAVE * KPT) + lane;

        // ---- load + digit + match: independent -> ILP ------------
        __builtin_amdgcn_s_setprio(0);
        if (!preloaded) load_tile<ES, KPT>(e, src + tile_base, seg, valid, full);
#ifdef RSX_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // diagnostic build: split load wait from match
        RSX_STAMP(7);
#endif
        if constexpr ((XF & 1) != 0) {  // first pass of a sort: keys become order-preserving unsigned
#pragma unroll
            for (int j = 0; j < KPT; ++j) key_map<ES, false>(e[j], a.xf);
        }
        // ---- rank within the wave (stable): for every element, the number of elements of this wave
        // with the same digit that come before it in (round, lane) order, plus the wave's running
        // count of that digit.  Two ways, chosen per tile and wave:
        //  * ballots: wave64 "match any" (the lanes sharing my digit, m) in 32 hand-scheduled VALU;
        //    every lane reads its digit's running count, the first lane of each group adds the group
        //    size with an LDS atomic (the read never feeds the write, so the LDS ops pipeline).
        //    Cost independent of the digit distribution.
        //  * LDS atomics: every lane adds 1 to its digit's counter and takes the returned old value as
        //    its rank.  One LDS instruction instead of ~40 VALU -- but lanes that share an address are
        //    serialised, and the ranks are stable only if the LDS applies them in ascending lane
        //    order.  gfx950 does (not an ISA promise): rsx_lds_order_kernel tests exactly that when a
        //    context first touches the device, and a.rank_atomic is set only if it passed.
        // Round 0 always goes by ballots; if it shows a digit shared by >= a.hot_lanes lanes (skewed
        // digits: a constant high byte, few distinct keys) the whole tile does, else the other rounds
        // use the atomics.  Kept per element: its 16-bit tile rank, two per VGPR; the digit is
        // re-derived from the element where needed (2 VALU) instead of kept.
        uint32_t pk[(KPT + 1) / 2];
        uint64_t crowd = 0;   // round 0: lanes whose digit is shared by >= a.hot_lanes lanes
        uint32_t digit0 = 0;  // round 0: this lane's digit
        auto rank_keys = [&](auto is_full, auto by_atomic, auto j_begin) __attribute__((always_inline)) -> bool {  // bodies duplicated per case
            constexpr bool FULL = decltype(is_full)::value;
            constexpr bool ATOM = decltype(by_atomic)::value;
            constexpr int JB = decltype(j_begin)::value;
            constexpr int JE = JB == 0 ? 1 : KPT;
            constexpr int RG = (JE - JB) < RSX_RANK_GROUP ? (JE - JB) : RSX_RANK_GROUP;
            bool crowded = false;
#pragma unroll
            for (int j0 = JB; j0 < JE; j0 += RG) {
                uint32_t below[RG], word[RG], sh[RG];
#pragma unroll
                for (int r = 0; r < RG; ++r) {
                    const int j = j0 + r;
                    if (j >= JE) continue;
                    uint32_t d = sweep_digit<ES, STR>(e[j], spec);
                    if constexpr (!FULL) {
                        if (seg + j * WAVE >= valid) d = 255u;
                    }
                    sh[r] = WIDE_CNT ? 0u : (d & 1u) * 16u;  // else two 16-bit counters per LDS word
                    uint32_t* cnt = &my_hist2[WIDE_CNT ? d : (d >> 1)];
                    if constexpr (ATOM) {
                        below[r] = 0;
                        word[r] = atomicAdd(cnt, 1u << sh[r]);
                        // RSX_OPT_RANK_CHECK (tests): the ordering the atomic ranks rest on, checked on REAL sweeps
                        // -- under whatever LDS contention the co-resident workgroups make -- not only by the
                        // idle-device self-test: within one instruction, the value returned to a lane must exceed
                        // the value returned to the first lane of its digit by the number of lower lanes with
                        // that digit.  One round per tile; a mismatch raises the context's error word.
                        if ((a.opts & SWEEP_OPT_RANK_CHECK) && j == 1) {
                            const uint64_t m = match_digit(d);
                            const uint32_t mine = (word[r] >> sh[r]) & (WIDE_CNT ? ~0u : 0xFFFFu);
                            const uint32_t first = (uint32_t)__shfl((int)mine, (int)__builtin_ctzll(m));
                            if (mine - first != mbcnt64(m)) __hip_atomic_store(a.error, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                    } else {
                        uint64_t m;
                        if constexpr (FULL && JB != 0) {
                            // crowded tiles (the only ones that come here past round 0) often have rounds
                            // where the whole wave holds ONE digit -- a constant byte, sorted input: no match needed
                            const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
                            if (__ballot(d != d0) == 0) m = ~0ull;
                            else m = match_digit_sched(d);
                        } else if constexpr (FULL) {
                            m = match_digit_sched(d);
                        } else {
                            m = match_digit(d);
                        }
                        below[r] = mbcnt64(m);
                        const uint32_t group = (uint32_t)__popcll(m);
                        word[r] = *cnt;
                        if (below[r] == 0) atomicAdd(cnt, group << sh[r]);
                        if constexpr (JB == 0) {
                            crowd = __ballot(group >= a.hot_lanes);
                            digit0 = d;
                            crowded = crowd != 0;
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < RG; ++r) {
                    const int j = j0 + r;
                    if (j >= JE) continue;
                    const uint32_t rank = (WIDE_CNT ? word[r] : ((word[r] >> sh[r]) & 0xFFFFu)) + below[r];
                    pk[j / 2] = (j & 1) ? (pk[j / 2] | (rank << 16)) : rank;
                }
            }
            return crowded;
        };
        // One dominant digit in the tile (a constant byte, sorted input, the head of a Zipf law): the lanes
        // holding it are ranked from a single ballot against a running count kept in a SCALAR (no LDS
        // traffic at all for them: same-address atomics are what skew makes expensive); the other lanes
        // use the atomics under their exec mask.  The scalar starts from what round 0 left in the wave's
        // counter and is added back to it once at the end.
        auto rank_hot = [&](auto is_full, uint32_t hotd) __attribute__((always_inline)) {
            constexpr bool FULL = decltype(is_full)::value;
            constexpr int RG = (KPT - 1) < RSX_RANK_GROUP ? (KPT - 1) : RSX_RANK_GROUP;
            const uint32_t hot_sh = WIDE_CNT ? 0u : (hotd & 1u) * 16u;
            uint32_t* hot_cnt = &my_hist2[WIDE_CNT ? hotd : (hotd >> 1)];
            const uint32_t hot_init = (uint32_t)__builtin_amdgcn_readfirstlane((int)((*hot_cnt >> hot_sh) & (WIDE_CNT ? ~0u : 0xFFFFu)));
            uint32_t hot_run = hot_init;  // wave-uniform
#pragma unroll
            for (int j0 = 1; j0 < KPT; j0 += RG) {
                uint32_t word[RG], sh[RG], below[RG];
#pragma unroll
                for (int r = 0; r < RG; ++r) {
                    const int j = j0 + r;
                    if (j >= KPT) continue;
                    uint32_t d = sweep_digit<ES, STR>(e[j], spec);
                    if constexpr (!FULL) {
                        if (seg + j * WAVE >= valid) d = 255u;
                    }
                    const uint64_t h = __ballot(d == hotd);
                    sh[r] = WIDE_CNT ? 0u : (d & 1u) * 16u;
                    if constexpr (ES <= 4) {
                        // every lane takes the hot value FIRST and the other lanes overwrite it with their atomic's
                        // return: written as a branch per case, the hot lanes' move lands in the register the atomic
                        // is still returning into, and every round waits for its LDS atomic (s_waitcnt lgkmcnt(0)
                        // per round).  Zipf u32 -1.7 %; 8-byte elements measured 2 % slower this way, so they keep
                        // the branch.
#if RSX_HOT_MBCNT
                        // ... and that hot value is already the lane's rank: v_mbcnt takes an addend, so
                        // hot_run + (hot lanes below me) is two VALU with the running count as the addend -- no select, no
                        // separate `below` (the other lanes' atomic return overwrites it)
                        uint32_t w = __builtin_amdgcn_mbcnt_hi((uint32_t)(h >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)h, hot_run)) << sh[r];
                        if (d != hotd) w = atomicAdd(&my_hist2[WIDE_CNT ? d : (d >> 1)], 1u << sh[r]);
                        word[r] = w;
                        below[r] = 0u;
#else
                        uint32_t w = hot_run << sh[r];
                        if (d != hotd) w = atomicAdd(&my_hist2[WIDE_CNT ? d : (d >> 1)], 1u << sh[r]);
                        word[r] = w;
                        below[r] = d == hotd ? mbcnt64(h) : 0u;
#endif
                    } else if (d == hotd) {
                        below[r] = mbcnt64(h);
                        word[r] = hot_run << sh[r];
                    } else {
                        below[r] = 0;
                        word[r] = atomicAdd(&my_hist2[WIDE_CNT ? d : (d >> 1)], 1u << sh[r]);
                    }
                    hot_run += (uint32_t)__popcll(h);
                }
#pragma unroll
                for (int r = 0; r < RG; ++r) {
                    const int j = j0 + r;
                    if (j >= KPT) continue;
                    const uint32_t rank = (WIDE_CNT ? word[r] : ((word[r] >> sh[r]) & 0xFFFFu)) + below[r];
                    pk[j / 2] = (j & 1) ? (pk[j / 2] | (rank << 16)) : rank;
                }
            }
            if (lane == 0 && hot_run != hot_init) atomicAdd(hot_cnt, (hot_run - hot_init) << hot_sh);
        };
        auto match_rank = [&](auto is_full) __attribute__((always_inline)) {
            const bool crowded = rank_keys(is_full, std::false_type{}, std::integral_constant<int, 0>{});
            if (a.rank_atomic && !crowded) {
                rank_keys(is_full, std::true_type{}, std::integral_constant<int, 1>{});
            } else if (a.rank_atomic) {
                // is every crowded lane of round 0 on ONE digit?
                const uint32_t hotd = (uint32_t)__builtin_amdgcn_readlane((int)digit0, (int)__builtin_ctzll(crowd));
                if ((crowd & ~__ballot(digit0 == hotd)) == 0) rank_hot(is_full, hotd);
                else rank_keys(is_full, std::false_type{}, std::integral_constant<int, 1>{});
            } else {
                rank_keys(is_full, std::false_type{}, std::integral_constant<int, 1>{});
            }
        };
        if (full) match_rank(std::true_type{});
        else match_rank(std::false_type{});
        RSX_STAMP(1);
        __syncthreads();
        RSX_STAMP(2);
        // from here to the end of the tile the workgroup is on short, serial phases that every
        // wave waits for: let them win issue arbitration over other workgroups' match phases
        if (!RSX_DBG(a, 0x1000u)) __builtin_amdgcn_s_setprio(2);

        // ---- per-digit: wave counts -> tile count, publish aggregate --------------
        uint32_t tcount = 0;
        if (tid < RADIX) {
#pragma unroll
            for (int w = 0; w < NWAVE; ++w) tcount += s_whist[w * RADIX + tid];
            const uint32_t real = (tid == 255) ? tcount - pad : tcount;
            const S flag = (kt == 0) ? (S)2 : (S)1;  // first tile of a chain: aggregate == inclusive
            publish(&status[stat_row + tid], (flag << Status<S>::SHIFT) | (S)real);
        }
        // the status words of the NEXT pass live in a second array; each tile zeroes its row there,
        // which spares a memset launch per pass
        if (a.status_clean != nullptr && tid < RADIX) static_cast<S*>(a.status_clean)[stat_row + tid] = 0;
        // ---- decoupled look-back inside the region's chain: state of digit `tid` (waves 0..3) ------
        // The walk back over the chain's earlier tiles is a series of dependent round trips (~500 cycles
        // each from the chain's L2).  Its first hop is requested now and looked at after the LDS reorder
        // (an aggregate or inclusive word stays true however old it is; an empty one is read again); the
        // rest of the walk follows the reorder.  RSX_LB_OVERLAP = 1 takes further steps of the walk BESIDE
        // the reorder (one after every RSX_LB_EVERY elements a wave has put into LDS) and publishes the
        // inclusive prefix the moment the walk ends: measured slower on 4-byte keys (256M u32: 0.468 ->
        // 0.485-0.498 ms per pass at every step spacing tried) and no different on 8- and 16-byte elements --
        // the waves that walk stall inside the reorder on each word, and the other four wait for them at
        // the barrier either way.
        constexpr int LBW = RSX_LB_WINDOW;  // words requested per round trip (the chain's earlier tiles, nearest first)
        S lb_pend[LBW];        // the words in flight
        uint32_t lb_n = 0;     // how many of them were requested
        S lb_excl = 0;         // sum of the predecessors' counts so far (region-relative: fits S)
        uint32_t lb_left = 0;  // predecessors not yet summed (tiles kt-1 .. 0 of the chain); 0 = walk finished
#pragma unroll
        for (int i = 0; i < LBW; ++i) lb_pend[i] = 0;
        // request up to `want` words starting with the nearest predecessor not yet summed
        auto lb_request = [&](uint32_t want) __attribute__((always_inline)) {
            const uint32_t dist = kt - lb_left + 1u;  // 1 = the tile right before this one
            lb_n = lb_left < want ? lb_left : want;
#pragma unroll
            for (int i = 0; i < LBW; ++i)
                if ((uint32_t)i < lb_n)
                    lb_pend[i] = __hip_atomic_load(&status[stat_row - (uint64_t)(dist + (uint32_t)i) * RADIX + tid], __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
        };
        if ((EARLY_HOP || RSX_LB_OVERLAP) && tid < RADIX && kt > 0) {
            lb_left = kt;
            lb_request(1u);  // early: only the nearest (the words behind it are worth more when read later)
        }
        // one step of the walk: take the words in flight (in order, up to the first empty one), request the next
        // window.  false = the first word was still empty.
        auto lb_step = [&]() __attribute__((always_inline)) -> bool {
            if (lb_left == 0) return true;
            bool progressed = false, stalled = false;
#pragma unroll
            for (int i = 0; i < LBW; ++i) {
                if ((uint32_t)i < lb_n && !stalled && lb_left != 0) {
                    const uint32_t f = (uint32_t)(lb_pend[i] >> Status<S>::SHIFT);
                    if (f != 0) {
                        lb_excl += (S)(lb_pend[i] & Status<S>::MASK);
                        lb_left = f == 2 ? 0u : lb_left - 1u;
                        progressed = true;
                    } else {
                        stalled = true;
                    }
                }
            }
            if (lb_left == 0) {
                const uint32_t real = (tid == 255) ? tcount - pad : tcount;
                publish(&status[stat_row + tid], ((S)2 << Status<S>::SHIFT) | (S)((lb_excl + (S)real) & Status<S>::MASK));
            } else {
                lb_request((uint32_t)LBW);
            }
            return progressed;
        };
        // exclusive scan of tcount over the 256 digits -> start of each digit's run in the tile
        uint32_t incl = tcount;
        if (tid < RADIX) {
            incl = wave_incl_scan<(RSX_DPP_SCAN == 1 || (RSX_DPP_SCAN == 2 && ES <= 4))>(incl);
            if (lane == 63) s_misc[12 + wave] = incl;
        }
        __syncthreads();
        uint32_t tstart = 0;
        if (tid < RADIX) {
            uint32_t wbase = 0;
            for (uint32_t w = 0; w < wave; ++w) wbase += s_misc[12 + w];
            tstart = wbase + incl - tcount;
            uint32_t run = tstart;
#pragma unroll
            for (int w = 0; w < NWAVE; ++w) {  // counts are re-read rather than kept in 8 VGPRs across the barrier
                const uint32_t c = s_whist[w * RADIX + tid];
                s_whist[w * RADIX + tid] = (Cnt)run;
                run += c;
            }
        }
        __syncthreads();
        RSX_STAMP(3);

        // ---- reorder the tile in LDS by digit -------------------------------------
        forget<ES, KPT>(e);
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const uint32_t d = (!full && seg + j * WAVE >= valid) ? 255u : sweep_digit<ES, STR>(e[j], spec);
            // bank swizzle (RSX_LDS_SWIZZLE): slot p lives at p ^ ((p >> 5) & 31).  Digit runs that start
            // a multiple of 32 slots apart -- every pass over already sorted input, key = index -- would
            // otherwise put all 64 lanes of a wave on one bank
            const uint32_t pos = my_hist[d] + ((pk[j / 2] >> (16 * (j & 1))) & 0xFFFFu);
            s_elems[RSX_LDS_SWIZZLE ? (pos ^ ((pos >> 5) & 31u)) : pos] = e[j];
            if (RSX_LB_OVERLAP && j % RSX_LB_EVERY == RSX_LB_EVERY - 1 && j + 1 < KPT && tid < RADIX) lb_step();  // (wave-uniform)
        }
        __syncthreads();  // s_whist is dead from here: s_base takes its place
        RSX_STAMP(4);
        // physical LDS slot of logical slot i*WG + tid (see the swizzle at the reorder): only the low
        // five bits change, by (i*WG/32 + tid/32) & 31 -- two values per thread when WG % 512 == 0
        static_assert(!RSX_LDS_SWIZZLE || WG % 512 == 0, "swizzle constants assume WG % 512 == 0");
        const uint32_t sw0 = RSX_LDS_SWIZZLE ? (tid ^ ((tid >> 5) & 31u)) : tid;
        const uint32_t sw1 = RSX_LDS_SWIZZLE ? (tid ^ (((tid >> 5) + (WG >> 5)) & 31u)) : tid;
        auto slot_of = [&](int i) __attribute__((always_inline)) { return (uint32_t)(i * WG) + ((i & 1) ? sw1 : sw0); };
        // The elements a thread will write out are read from LDS NOW, ahead of the look-back: the reads are in
        // flight while the digit threads walk the chain and the other waves would only wait at the barrier.
        constexpr int PREREAD_WANT = ES <= 4 ? RSX_WO_PREREAD : ES == 8 ? RSX_WO_PREREAD8 : 0;
        constexpr int PREREAD = PREFETCH ? 0 : (PREREAD_WANT < KPT ? PREREAD_WANT : KPT);  // elements read ahead
        // (256M u32 -2.3 %, 512M -0.7 %, 1B +1.0 %: the host switches it on up to 2 GiB of data; 8-byte elements lose at
        // any count.)
        const bool preread = PREREAD > 0 && (a.opts & SWEEP_OPT_PREREAD) != 0u;  // wave-uniform
        if constexpr (PREREAD > 0) {
            if (full && preread) {
#pragma unroll
                for (int i = 0; i < PREREAD; ++i) e[i] = s_elems[slot_of(i)];
            }
        }
        // static mode knows its next tile: its loads are issued as soon as the element registers
        // are free and fly during the look-back and the write-out.  The waves that do the look-back
        // issue theirs AFTER it: memory operations return in order, so a status word requested
        // behind 16 HBM loads would wait for all of them.
        const bool prefetch = static_mode && PREFETCH && st_k < st_nt;
        auto issue_next = [&]() {
            const uint64_t tb = ((uint64_t)home << a.g.region_shift) + (uint64_t)st_k * TILE;
            const uint64_t re = ((uint64_t)(home + 1) << a.g.region_shift) < a.g.n ? ((uint64_t)(home + 1) << a.g.region_shift) : a.g.n;
            const bool fl = re - tb >= (uint64_t)TILE;
            load_tile<ES, KPT>(e, src + tb, seg, fl ? (uint32_t)TILE : (uint32_t)(re - tb), fl);
        };
        preloaded = prefetch;
        if (prefetch && RSX_PREFETCH_ALL == 1 && tid >= RADIX) issue_next();

        // ---- what is left of the look-back ------------------------------------------------------
        if (tid < RADIX) {
            if (!(EARLY_HOP || RSX_LB_OVERLAP) && kt > 0) {  // nothing requested yet
                lb_left = kt;
                lb_request((uint32_t)LBW);
            }
            uint32_t spins = 0;
#ifdef RSX_STAMPS
            const unsigned long long lbs_t0 = __builtin_amdgcn_s_memtime();
            if (tid == 0 && kt > 0) ++lbs_tiles;
#endif
            while (lb_left != 0) {
#ifdef RSX_STAMPS
                if (tid == 0) ++lbs_hops;
#endif
                if (!lb_step()) {
#ifdef RSX_STAMPS
                    if (tid == 0) ++lbs_spins;
#endif
                    if (++spins > (1u << 22)) {  // bounded: never hang the device
                        __hip_atomic_store(a.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
#ifdef RSX_STAMPS
            if (tid == 0 && kt > 0) lbs_cycles += __builtin_amdgcn_s_memtime() - lbs_t0;
#endif
            // element index of LDS slot 0 if it belonged to this digit's run (wrap-safe in u64)
            s_base[tid] = rbase + (uint64_t)lb_excl - (uint64_t)tstart;
            if (prefetch && RSX_PREFETCH_ALL == 1) issue_next();
        }
        __syncthreads();
        RSX_STAMP(5);
        if (prefetch && RSX_PREFETCH_ALL >= 2) issue_next();  // all waves, behind the look-back: flies during the write-out

        // ---- write runs: consecutive threads -> consecutive addresses within a run;
        // ---- count the NEXT pass's digit per destination region on the way out
        if (!RSX_DBG(a, 2u)) {
            E* __restrict__ dst = static_cast<E*>(a.dst);
            // bin of the next pass's count matrix: (destination region, next digit)
            auto next_bin = [&](uint64_t idx, const E& x) -> uint32_t {
                uint32_t r;
                if constexpr (sizeof(S) == 4)  // 32-bit status words <=> region_shift <= 30: one funnel shift
                    r = __builtin_amdgcn_alignbit((uint32_t)(idx >> 32), (uint32_t)idx, a.g.region_shift);
                else
                    r = (uint32_t)(idx >> a.g.region_shift);
                return (r << 8) | sweep_digit<ES, STR>(x, next);
            };
            if (full) {
                auto write_full = [&](auto crowd, uint32_t hot_nd) __attribute__((always_inline)) {  // duplicated: plain LDS atomics / skew-proof counting
                    constexpr bool CROWD = decltype(crowd)::value;
                    // CROWD: the hot next digit's elements are counted in a SCALAR per wave (one ballot and one
                    // population count per instruction, no LDS traffic), keyed by the bin hot_bin = (region, hot
                    // digit); the scalar is flushed when the wave's slots move on to another region, and at the
                    // end of the tile.  Lanes on other bins add 1 each under their exec mask.
                    uint32_t hot_bin = ~0u, hot_acc = 0;
                    auto hot_flush = [&]() __attribute__((always_inline)) {
                        if (hot_acc != 0 && lane == 0) atomicAdd(&s_jn[hot_bin], hot_acc);
                        hot_acc = 0;
                    };
#pragma unroll
                    for (int i = 0; i < KPT; ++i) {
                        const uint32_t p = i * WG + tid;
                        E x;
                        if (i < PREREAD && preread) x = e[i];
                        else x = s_elems[slot_of(i)];
                        const uint64_t idx = s_base[sweep_digit<ES, STR>(x, spec)] + p;
                        if constexpr ((XF & 2) != 0) {  // last pass: back to the caller's representation
                            E y = x;
                            key_map<ES, true>(y, a.xf);
                            dst[idx] = y;
                        } else {
                            dst[idx] = x;
                        }
                        if constexpr (NEXT) {
                          if (!MID || do_next) {
                            if constexpr (CROWD) {
                                const uint32_t bin = next_bin(idx, x);
                                uint64_t same = __ballot(bin == hot_bin);
                                if (same == 0) {  // wave-uniform: no lane on the remembered bin -- the wave's slots have moved on
                                    hot_flush();
                                    // to another destination region: the tile's hot digit there (the head of a Zipf law) ...
                                    hot_bin = ((uint32_t)__builtin_amdgcn_readfirstlane((int)(bin >> 8)) << 8) | hot_nd;
                                    same = __ballot(bin == hot_bin);
                                    if (same == 0) {  // ... or to another run of equal keys (few distinct values): the first lane's bin
                                        hot_bin = (uint32_t)__builtin_amdgcn_readfirstlane((int)bin);
                                        same = __ballot(bin == hot_bin);
                                    }
                                }
                                hot_acc += (uint32_t)__popcll(same);  // (a per-LANE counter summed at the flushes: Zipf u32 +3.5 %, step-16 +2.6 %)
                                if (bin != hot_bin) atomicAdd(&s_jn[bin], 1u);
                            } else if (!RSX_DBG(a, 0x8u)) {
                                atomicAdd(&s_jn[next_bin(idx, x)], 1u);  // (0x8: ablation, no next-pass count)
                            }
                          }
                        }
                        // (element by element on purpose: issuing a group's LDS reads ahead of its atomics was
                        // measured no faster on u32 and slower on 8/16-byte elements)
                        if (i % RSX_WO_GROUP == RSX_WO_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (NEXT && CROWD) hot_flush();
                };
                // Same-address LDS atomics serialise.  One probe per wave and tile: if a quarter of the
                // wave's first 64 elements share their next digit, count the careful way (per-element
                // check, wave match on crowded bins); else one plain atomic per element.
                bool crowded_next = false;
                uint32_t hot_nd = 0;
                if (do_next) {
                    const uint32_t nd = sweep_digit<ES, STR>(s_elems[sw0], next);
                    hot_nd = (uint32_t)__builtin_amdgcn_readfirstlane((int)nd);
                    crowded_next = __popcll(__ballot(nd == hot_nd)) >= 16;
                }
                if (crowded_next) write_full(std::true_type{}, hot_nd);
                else write_full(std::false_type{}, 0u);
            } else {
#pragma unroll
                for (int i = 0; i < KPT; ++i) {
                    const uint32_t p = i * WG + tid;
                    if (p < valid) {
                        const E x = s_elems[slot_of(i)];
                        const uint64_t idx = s_base[sweep_digit<ES, STR>(x, spec)] + p;
                        if constexpr ((XF & 2) != 0) {
                            E y = x;
                            key_map<ES, true>(y, a.xf);
                            dst[idx] = y;
                        } else {
                            dst[idx] = x;
                        }
                        if (do_next)
                            count_next(s_jn, next_bin(idx, x));
                    }
                }
            }
        }
        __syncthreads();  // s_elems / s_base are reused by the next tile
        RSX_STAMP(6);
    }

    if (do_next) {  // hand this workgroup's share of the next count matrix over
        for (uint32_t i = threadIdx.x; i < a.g.num_regions * RADIX; i += WG) {
            const uint32_t c = s_jn[i];
            if (c)
                __hip_atomic_fetch_add(reinterpret_cast<JT*>(a.jnext) + (blockIdx.x % J_REPL) * (a.g.num_regions * RADIX) + i, (JT)c,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#ifdef RSX_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long stamp_end = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 8; ++k) atomicAdd(&a.dbg_cnt[(threadIdx.x >> 6) * 8 + k], stamp_acc[k]);
    if (threadIdx.x == 0) {  // prologue / epilogue of the workgroup: [64] roll call, [65] cursors, [66] flush, [67] workgroups
        atomicAdd(&a.dbg_cnt[64], stamp_called - stamp_entry);
        atomicAdd(&a.dbg_cnt[65], stamp_loop - stamp_called);
        atomicAdd(&a.dbg_cnt[66], stamp_end - stamp_prev);
        atomicAdd(&a.dbg_cnt[67], 1ull);
        atomicAdd(&a.dbg_cnt[68], lbs_hops);    // look-back of digit 0: words looked at (the early hop excluded when it was ready),
        atomicAdd(&a.dbg_cnt[69], lbs_spins);   // of which still empty,
        atomicAdd(&a.dbg_cnt[70], lbs_cycles);  // cycles in the walk after the reorder,
        atomicAdd(&a.dbg_cnt[71], lbs_tiles);   // tiles with a predecessor
    }
#endif
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

}  // namespace rsx
