// rsx_internal.hpp -- host-side state and helpers shared by the translation units of librsx.so:
//   rsx.hip     the C-ABI of include/rsx.h (context, pass loop, multi-GPU driver, harness)
//   rsx_es.hip  the kernel launchers of ONE element size (compiled once per size with -DRSX_ES=n,
//               so the eight sizes build in parallel)
#pragma once
#include "rsx_device.hpp"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <algorithm>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/rsx.h"

namespace rsxh {
using namespace rsx;

// aux block layout (one hipMalloc, zeroed at creation).
// What a sort ACCUMULATES into before any of its kernels could clear it -- every pass's ticket / roll-call words, the
// count matrix of the most significant digit (middle sizes) and count matrix 0 -- lives in a CONTROL BLOCK, and there
// are three of them: sort k uses block k % 2 and its count kernel, on its way, zeroes the block sort k - 1 used, so
// that no memset launch stands between two sorts (3.5 us of 60 at 2^16 keys).  Block 2 belongs to sorts that are being
// captured into a graph: a replay cannot alternate, so those zero their block themselves, with a memset, as every
// sort did before.  Count matrices 1 and 2 are cleared by the kernels of the sort itself (count kernel, first sweep).
// A count matrix is kept in J_REPL replicas (rsx_device.hpp): [J_REPL][num_regions][256] u64.
constexpr int MAX_PASSES = 16;                                                   // u128 keys
constexpr size_t J_BYTES = (size_t)J_REPL * MAX_REGIONS * RADIX * sizeof(uint64_t);  // one count matrix, all replicas
constexpr size_t TICKET_WORDS = ROLL_SHARDS + (size_t)ROLL_SHARD_COUNT * ROLL_SHARD_STRIDE;  // per pass (rsx_device.hpp)
constexpr uint32_t PART_MAX_SUB = 16;  // sub-ranges of rsx_partition_count_device / rsx_partition_scatter_device
constexpr uint32_t MID_MAX_REGIONS = 16;  // middle-size sorts: regions of the most significant digit's count matrix
constexpr size_t CB_TICKETS = 0;                                                 // [MAX_PASSES][TICKET_WORDS] u32
constexpr size_t CB_JT = ((MAX_PASSES * TICKET_WORDS * 4 + 255) / 256) * 256;    // count matrix of the most significant digit
constexpr size_t JT_BYTES = (size_t)J_REPL * MID_MAX_REGIONS * RADIX * sizeof(uint64_t);
constexpr size_t CB_J0 = CB_JT + JT_BYTES;                                       // count matrix 0
constexpr size_t CB_BYTES = CB_J0 + J_BYTES;
static_assert(CB_BYTES % 16 == 0, "control blocks are zeroed with 16-byte stores");
constexpr int CB_COUNT = 3;
constexpr size_t OFF_J1 = CB_COUNT * CB_BYTES;
constexpr size_t OFF_J2 = OFF_J1 + J_BYTES;
constexpr size_t OFF_PART_TICKETS = OFF_J2 + J_BYTES;                            // control words of rsx_partition_scatter_device
constexpr size_t OFF_BASE = OFF_PART_TICKETS + ((TICKET_WORDS * 4 + 255) / 256) * 256;  // scratch of the context self-tests
constexpr size_t OFF_FLAGS = OFF_BASE + 4096;                                    // self-test verdicts
constexpr size_t OFF_DBG = OFF_FLAGS + 256;                                      // 16 waves x 8 diagnostic counters
constexpr size_t AUX_BYTES = OFF_DBG + 1024;

// option bits (rsx_ctx_set_option): alternative kernel paths, all bit-exact
enum : uint32_t {
    OPT_DYNAMIC_TILES = 1u << 0,   // ticketed tiles, no roll call
    OPT_BALLOT_RANKS = 1u << 1,    // never rank by returned LDS atomics
    OPT_ATOMIC_RANKS = 1u << 2,    // atomics whatever the skew
    OPT_AGENT_STATUS = 1u << 3,    // agent-scope status stores everywhere
    OPT_NO_XCD_MAJOR = 1u << 4,    // plain blockIdx numbering
    OPT_GENERAL_BYTES = 1u << 5,   // one-byte elements through the general pass
    OPT_VERBOSE = 1u << 6,
    OPT_RANK_CHECK = 1u << 7,      // cross-check atomic ranks against ballots on real tiles (tests)
    OPT_NO_SMALL_SORT = 1u << 8,   // arrays of at most one tile through the general path too
    OPT_NO_MID_SORT = 1u << 9,     // middle sizes through the general path too (no bucket split)
};

}  // namespace rsxh

struct rsx_ctx {
    int device = 0;
    std::mutex mu;
    std::string err = "";
    void* status = nullptr;  // tile status words: two halves, alternating per pass
    size_t status_bytes = 0;
    char* aux = nullptr;
    uint32_t* host_err = nullptr;  // pinned, device-mapped: a kernel that gives up sets it (no sync needed to see it)
    uint32_t* host_err_dev = nullptr;
    // staging for rsx_sort_host
    void* host_buf[2] = {nullptr, nullptr};
    size_t host_bytes = 0;
    hipStream_t copy_stream[2] = {nullptr, nullptr};
    void* pinned[4] = {nullptr, nullptr, nullptr, nullptr};  // ring of pinned bounce chunks
    hipEvent_t copy_event[4] = {nullptr, nullptr, nullptr, nullptr};
    // multi-GPU driver (rsx_sort_sharded): per-slice stream and splitter-search scratch, made once
    rsx::Gate gate = {nullptr, 0, 0};  // set around the launches of a gated kernel sequence (wide keys)
    uint32_t wide_skip = 0;     // sorts to go without trying the wide-key hybrid (the last try was refused on the device)
    char* wide_buf = nullptr;   // wide-key hybrid: bin totals [65536] u64, bin-block sums [256] u64, bucket starts [65537] u64, verdict u32
    uint32_t wide_mode = 1;     // RSX_OPT_WIDE_SORT: 0 off, 1 auto, 2 always, 3 auto without the size floor
    std::vector<const void*> lds_attr;  // kernels whose dynamic-LDS limit was raised on this context's device (ensure_lds)
    const rsx::DigitSpec* spec_dev = nullptr;  // the hybrid's sweeps read their digits from the device's plan (WidePlan::specs)
    uint64_t wide_tried_sig = 0, wide_refused_sig = 0;  // (layout, n) of the last hybrid try / of the last refusal
    uint32_t bucket_no_skip = 0;  // RSX_OPT_BUCKET_SKIP == 0
    uint32_t bucket_group = 1;    // RSX_OPT_BUCKET_GROUP: small buckets of the hybrid are sorted in groups
    uint32_t* ovf16 = nullptr;  // u16 / i16 counting path: 65536 overflow counters, all zero between sorts
    unsigned long long* part_J = nullptr;  // rsx_partition_count_device: one count matrix per sub-range (PART_MAX_SUB x J_BYTES)
    hipStream_t shard_stream = nullptr;
    uint64_t* shard_q = nullptr;     // device: queries (lo, hi) + ranges (begin, end)
    uint64_t* shard_out = nullptr;   // device: answers
    uint64_t* shard_hist = nullptr;  // device: 256 digit counts
    uint64_t* shard_host = nullptr;  // pinned: answers, then the 256 counts
    std::vector<uint64_t> shard_stage;
    int num_cu = 256;
    uint32_t pass_index = 0;   // of the sweep being launched within its sort (selects the status half, J rotation)
    bool pass_last = true;     // no pass follows: nothing to clean
    uint32_t last_path = 0;    // 0 general passes, 1 one-launch sort, 2 middle-size bucket split, 3 / 4 one- / two-byte counting
    uint32_t last_sort_passes = 0;  // sweep passes of the last sort (RSX_INFO_LAST_PASSES)
    rsx::CleanList clean = {{nullptr, nullptr, nullptr}, {0, 0, 0}};  // what the next count kernel zeroes on its way (the previous sort's control block)
    uint64_t cb_used[2][2] = {{0, 0}, {0, 0}};  // per alternating block: bytes of the top-digit matrix / of count matrix 0 its last sort used
    uint32_t cb_alt = 0;       // which of the two alternating blocks the last uncaptured sort used
    uint32_t* tickets_override = nullptr;  // rsx_partition_scatter_device: control words outside the blocks
    uint32_t cb = 0;           // control block of the sort being enqueued (aux layout above)
    uint32_t cb_last = 0;      // ... of the last sort that ran sweeps (RSX_INFO_LAST_PASSES)
    bool cb_dirty = false;     // an enqueue failed half way: both alternating blocks are zeroed by memset before the next sort
    uint32_t pass_mid = 0;     // middle-size sort, first sweep (MID instantiation): 1 = bucket split by the top digit, 2 = first LSD pass
    uint32_t mid_choice = 0;   // what the last middle-size sort was enqueued as (1 / 2)
    bool bucket_small = false; // the bucket kernel being launched: 256-thread workgroups
    uint32_t mid_force = 0;    // RSX_OPT_MID_SORT 2 / 3: always split (1) / always LSD passes (2)
    uint32_t mid_cooldown = 0; // sorts to go by LSD passes after a bucket split met a skewed input
    bool rank_atomic = false;  // LDS atomic ordering self-test passed (set when the workspace is first made)
    bool l2_local = false;     // same-XCD hand-off self-test passed: chains may keep status words in their L2
    uint32_t hot_lanes = 16;
    uint32_t options = 0;      // OPT_* (rsx_ctx_set_option)
    uint32_t max_regions = 0;  // 0 = default per element size
    uint32_t dbg = 0;          // RSX_TUNING builds only: timing ablations (wrong output by design)
    // stream-ordered reuse: work of this context on another stream waits for the last enqueue
    hipStream_t last_stream = nullptr;
    hipEvent_t last_event = nullptr;
    bool busy = false;
    // per-launch HIP-event timing (rsx_ctx_profile)
    bool prof = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pending[RSX_PROF_KINDS];
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_free;
    double prof_ms[RSX_PROF_KINDS] = {0, 0, 0, 0};
    uint64_t prof_n[RSX_PROF_KINDS] = {0, 0, 0, 0};
    std::vector<float> prof_each;  // per-launch sweep times (rsx_debug_sweep_times)
};

namespace rsxh {

inline int fail(rsx_ctx* c, int code, const char* what, hipError_t e = hipSuccess) {
    if (c) {
        c->err = what;
        if (e != hipSuccess) {
            c->err += ": ";
            c->err += hipGetErrorString(e);
        }
    }
    return code;
}

#define RSX_HIP(call)                                                   \
    do {                                                                \
        hipError_t _e = (call);                                         \
        if (_e != hipSuccess) return fail(ctx, RSX_ERR_HIP, #call, _e); \
    } while (0)

// Records a start/stop event pair around one launch when profiling is on.
struct LaunchTimer {
    rsx_ctx* c;
    int kind;
    hipStream_t st;
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    LaunchTimer(rsx_ctx* ctx, int k, hipStream_t s) : c(ctx), kind(k), st(s) {
        if (!c->prof) return;
        if (!c->prof_free.empty()) {
            ev = c->prof_free.back();
            c->prof_free.pop_back();
        } else if (hipEventCreate(&ev.first) != hipSuccess || hipEventCreate(&ev.second) != hipSuccess) {
            ev = {nullptr, nullptr};
            return;
        }
        (void)hipEventRecord(ev.first, st);
    }
    ~LaunchTimer() {
        if (!ev.first) return;
        (void)hipEventRecord(ev.second, st);
        c->prof_pending[kind].push_back(ev);
    }
};

// Keys per thread by element size.  Tiles need not be powers of two (the last tile of a region is
// partial anyway); bigger tiles mean longer output runs per digit (fewer partial cache lines, the
// memory system's real cost here) and fewer look-backs per key, as long as two or three workgroups
// still fit a CU: u32 28 x 512 = 14336 keys (56 KiB), u64 14 x 512 (56 KiB), 16-byte 5 x 512 (40 KiB),
// 12-byte 10 x 512 (60 KiB), 24-byte 5 x 512 (60 KiB), 32-byte 3 x 512 (48 KiB).
// Measured against 16 / 8 / 4: 1B u32 117 -> 136, 1B u64 32 -> 34.9, 128M (u64,u64) 17.3 -> 18 Gkeys/s.
#ifndef RSX_REGION_FLOOR
#define RSX_REGION_FLOOR 6  // log2 of the fewest tiles worth a region of their own
#endif
#ifndef RSX_KPT4
#define RSX_KPT4 28
#endif
#ifndef RSX_KPT2
#define RSX_KPT2 RSX_KPT4  // 1- and 2-byte elements
#endif
#ifndef RSX_WG4
#define RSX_WG4 512
#endif
#ifndef RSX_KPT8
#define RSX_KPT8 14  // 56 KiB tiles, still two workgroups per CU with 16 regions: 1B u64 30.8 -> 30.15 ms, Zipf u64 -3.4 % (12: round 1)
#endif
#ifndef RSX_KPT16
#define RSX_KPT16 5
#endif
#ifndef RSX_KPT12
#define RSX_KPT12 10
#endif
#ifndef RSX_KPT32
#define RSX_KPT32 3
#endif
#ifndef RSX_KPT24
#define RSX_KPT24 5  // with 8 regions: 60 KiB tiles at two workgroups per CU: 2 GiB of (u64,[u64;2]) 8.34 -> 7.53 ms (3 x 512, 16 regions)
#endif
#ifndef RSX_WG8
#define RSX_WG8 512
#endif
// rsx_bucket_sort_kernel: 1024 threads x BKPT elements in registers, the bucket in LDS (<= 112 KiB, 8-byte elements
// 136 KiB) beside 16 KiB of wave counters
constexpr int bucket_kpt_for(int es) { return es <= 4 ? 28 : es == 8 ? 17 : es == 12 ? 9 : es == 16 ? 7 : es == 24 ? 4 : 3; }
// ... and of the hybrid's 1024-thread form, which fills the LDS: as many registers as hold what 160 KiB leave beside the
// wave counters (bucket_cape: 16-byte elements 9020 of 9216 slots); its smaller forms and the middle sizes keep the shorter
// unrolled loops.
constexpr int wide_kpt_for(int es) { return es <= 4 ? 28 : es == 8 ? 17 : es == 12 ? 12 : es == 16 ? 9 : es == 24 ? 6 : 5; }
// ... and of rsx_bucket16_medium_kernel (1024 threads).  It inlines the split, the LDS sort and the sort through memory: with
// the bucket kernel's 17 eight-byte elements per thread it spilled 113-141 registers into 950-970 bytes of scratch per lane
// (16 per thread: 792 bytes; 15: 32), and a kernel with that much scratch takes 20-25 us to DISPATCH even when its gate sends
// it home at once -- every hybrid sort of 8-byte elements paid that (6 % of a 2^23-key sort).
#ifndef RSX_KMED8
#define RSX_KMED8 15
#endif
constexpr int medium_kpt_for(int es) { return es == 8 ? RSX_KMED8 : wide_kpt_for(es); }
// whether an array's 1024-thread form is the longer one: when the average bucket is above 7/8 of what the shorter holds
constexpr bool wide_big_form(int es, uint64_t n) { return n / 65536u > (uint64_t)1024 * (uint64_t)bucket_kpt_for(es) * 7u / 8u; }
// elements a workgroup of `wg` threads x `kpt` registers holds in LDS (rsx_small_kernel.hpp cape<ES, KPT, WG>, same formula)
constexpr uint32_t bucket_cape(int es, int kpt, int wg) {
    const uint32_t slots = (uint32_t)wg * (uint32_t)kpt;
    const uint32_t room = ((163840u - 1024u - (uint32_t)(wg / 64) * 256u * 4u - 64u - 3u * 256u * 4u) / (uint32_t)es) & ~3u;
    return slots < room ? slots : room;
}
constexpr size_t bucket_cnt_bytes(int) { return 4; }
// the hybrid's buffer (ctx->wide_buf): bucket totals [65536], block totals [256], starts [65537] (u64), then its WidePlan
constexpr size_t WIDE_PLAN_OFFSET = (65536 + 256 + 65537 + 1) * sizeof(uint64_t);
constexpr uint32_t bucket_cap(int es) { return 1024u * (uint32_t)bucket_kpt_for(es); }
// Largest array taken by the middle-size path: the average bucket is 4/7 of the capacity, so uniform top digits
// pass with a wide margin (2^22 4-byte, 2^20 16-byte elements); skewed ones fall back to LSD passes.  8-byte elements
// go up to 2^22 as well (the general path needs 275 us there, this one 90): the average bucket is then 16384 of 17408,
// eight standard deviations of a uniform top digit below the capacity; a bucket that overflows all the same is sorted
// through memory by its workgroup, which costs about what the general path would have.
constexpr uint64_t mid_max_elems(int es) { return es == 8 ? (1ull << 22) : (uint64_t)bucket_cap(es) * 256u * 4u / 7u; }
constexpr int kpt_for(int es) { return es <= 2 ? RSX_KPT2 : es <= 4 ? RSX_KPT4 : es == 8 ? RSX_KPT8 : es == 12 ? RSX_KPT12 : es == 16 ? RSX_KPT16 : es == 24 ? RSX_KPT24 : RSX_KPT32; }
constexpr int wg_for(int es) { return es <= 4 ? RSX_WG4 : es == 8 ? RSX_WG8 : 512; }
constexpr uint32_t tile_elems(int es) { return wg_for(es) * kpt_for(es); }
// The bucket split of a middle-size sort (rsx_mid_kernels.hpp) works on SMALL tiles, one workgroup each (a tile is one
// workgroup's serial work: 13 us for 14336 u32 keys, and 2^16 keys are five of those): 512 x 8 4-byte, 512 x 4 8-byte,
// 512 x 2 16-byte elements.
constexpr int mid_kpt_for(int es) { return es <= 4 ? 8 : es == 8 ? 4 : es == 12 ? 3 : 2; }
constexpr uint32_t mid_tile_elems(int es) { return 512u * (uint32_t)mid_kpt_for(es); }

inline uint32_t log2u(uint64_t x) { return 63u - (uint32_t)__builtin_clzll(x); }

// Regions: smallest power-of-two length (>= one tile) that covers n with <= cap of them.
// small_tiles: tile count of a middle-size sort's bucket split (mid_tile_elems): only its status_rows() is used, to size
// the workspace that holds the split's two tiles x 256 tables.
inline RegionGeom make_geom(const rsx_ctx* ctx, uint64_t n, uint32_t es, bool small_tiles = false) {
    RegionGeom g;
    g.n = n;
    g.tile = small_tiles ? mid_tile_elems((int)es) : tile_elems((int)es);
    uint32_t k = log2u(g.tile);
    if ((1ull << k) < g.tile) ++k;  // tiles need not be a power of two; regions are
    // A region is worth having from about 64 tiles on: every region ends with a partial tile and a workgroup of its
    // own, and widens every workgroup's count-matrix flush.  Measured with one region per 2^k keys: 2^18 u32
    // 109 -> 87 us, 2^22 u32 148 -> 134 us, 2^18 u64 249 -> 166 us, 2^22 u64 360 -> 275 us; large inputs are
    // bounded by `cap` as before.
    k += small_tiles ? 3 : RSX_REGION_FLOOR;
    // the next pass's count matrix costs 1 KiB of LDS per region: 8 where the tile needs the room
    const uint64_t cap = small_tiles ? MID_MAX_REGIONS : ctx->max_regions ? ctx->max_regions : (es == 8 || es == 32) ? 16 : 8;
    while (((n + (1ull << k) - 1) >> k) > cap) ++k;
    g.region_shift = k;
    g.num_regions = (uint32_t)((n + (1ull << k) - 1) >> k);
    if (g.num_regions == 0) g.num_regions = 1;
    return g;
}
inline uint64_t tiles_per_region(const RegionGeom& g, uint32_t) {
    return ((1ull << g.region_shift) + g.tile - 1) / g.tile;
}
inline uint64_t status_rows(const RegionGeom& g, uint32_t es) {
    return (uint64_t)g.num_regions * tiles_per_region(g, es);
}
// chain prefixes are relative to the region: 30 value bits suffice up to 2^30-element regions
inline bool status32(const RegionGeom& g) { return g.region_shift <= 30; }

// More than 64 KiB of dynamic LDS has to be asked for, per kernel and per DEVICE: once per context.
inline void ensure_lds(rsx_ctx* ctx, const void* kernel, size_t bytes) {
    for (const void* k : ctx->lds_attr)
        if (k == kernel) return;
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    ctx->lds_attr.push_back(kernel);
}

inline DigitSpec make_spec(const rsx_layout* L, uint32_t digit) {
    DigitSpec s;
    const uint32_t byte = L->key_offset + digit;
    const uint32_t top = L->key_offset + L->key_bytes - 1;
    if (L->elem_bytes >= 4) {
        s.word = byte >> 2;
        s.shift = 8 * (byte & 3);
        s.top_word = top >> 2;
        s.top_shift = 8 * (top & 3) + 7;
    } else {  // 1- and 2-byte elements live in one register
        s.word = 0;
        s.shift = 8 * byte;
        s.top_word = 0;
        s.top_shift = 8 * top + 7;
    }
    s.flip = (L->key_kind != RSX_KEY_UNSIGNED && digit == L->key_bytes - 1) ? 0x80u : 0u;
    s.fsign = L->key_kind == RSX_KEY_FLOAT ? ~0u : 0u;
    return s;
}

// the three count matrices rotate: pass d reads J_of(d % 3), accumulates the next pass's into
// J_of((d + 1) % 3) and zeroes J_of((d + 2) % 3) for the pass after
inline char* cb_of(rsx_ctx* c, uint32_t which) { return c->aux + (size_t)which * CB_BYTES; }
inline unsigned long long* J_of(rsx_ctx* c, uint32_t which) {
    char* p = which == 0 ? cb_of(c, c->cb) + CB_J0 : which == 1 ? c->aux + OFF_J1 : c->aux + OFF_J2;
    return reinterpret_cast<unsigned long long*>(p);
}
inline unsigned long long* JT_of(rsx_ctx* c) { return reinterpret_cast<unsigned long long*>(cb_of(c, c->cb) + CB_JT); }
inline uint64_t* base_of(rsx_ctx* c) { return reinterpret_cast<uint64_t*>(c->aux + OFF_BASE); }
inline uint32_t* tickets_of(rsx_ctx* c, uint32_t pass) {
    return reinterpret_cast<uint32_t*>(cb_of(c, c->cb) + CB_TICKETS) + (size_t)pass * TICKET_WORDS;
}
inline uint32_t* part_tickets_of(rsx_ctx* c) { return reinterpret_cast<uint32_t*>(c->aux + OFF_PART_TICKETS); }
inline uint32_t* flags_of(rsx_ctx* c) { return reinterpret_cast<uint32_t*>(c->aux + OFF_FLAGS); }

// ---- per-element-size launchers (defined in rsx_launch_impl.hpp, instantiated in rsx_es.hip) ----
// count phase of a first pass: J[rep][r][v] for `digit` over the input regions (J zeroed by the caller);
// jclear: a second count matrix to clear on the way (or null); clear_status: zero the tile status words of
// the sweep that follows (first half of the workspace)
template <int ES>
int launch_hist(rsx_ctx* ctx, const void* src, const RegionGeom& g, const rsx_layout* L, uint32_t digit,
                unsigned long long* J, unsigned long long* jclear, bool clear_status, hipStream_t st);
// the same counting a second digit (`digit2` into `J2`) on the same read (middle-size path)
template <int ES>
int launch_hist2(rsx_ctx* ctx, const void* src, const RegionGeom& g, const rsx_layout* L, uint32_t digit,
                 unsigned long long* J, uint32_t digit2, unsigned long long* J2, unsigned long long* jclear, hipStream_t st);
// first half of a middle-size sort: stable split of `src` into `dst` by the most significant digit (three launches)
template <int ES>
int launch_mid_split(rsx_ctx* ctx, const void* src, void* dst, size_t n, const rsx_layout* L, hipStream_t st);
// wide keys, large arrays: the top 16 bits of the mapped key counted per workgroup (P[parts][32768]); the 65536
// buckets (starts[65537]) sorted by the lower digits in LDS, in place
template <int ES>
int launch_wideplan(rsx_ctx* ctx, const void* src, size_t n, const rsx_layout* L, WidePlan* plan, hipStream_t st);
template <int ES>
int launch_count16top(rsx_ctx* ctx, const void* src, size_t n, const rsx_layout* L, WidePlan* plan, uint32_t* P, uint32_t parts,
                      uint32_t region_shift, uint32_t k, hipStream_t st);
template <int ES>
int launch_marginal16(rsx_ctx* ctx, const uint32_t* P, uint32_t parts, uint32_t k, const RegionGeom& g, unsigned long long* J,
                      unsigned long long* jclear, hipStream_t st);
template <int ES>
int launch_bucket16(rsx_ctx* ctx, void* data, void* scratch, size_t n, const rsx_layout* L, const uint64_t* starts, const WidePlan* plan,
                    hipStream_t st);
// second half of a middle-size sort: the 256 top-digit buckets of `src` sorted by the lower digits into `dst`
template <int ES>
int launch_bucket_sort(rsx_ctx* ctx, const void* src, void* dst, const RegionGeom& g, const rsx_layout* L, hipStream_t st);
// one sweep pass.  J: this pass's count matrix; jnext: accumulated for the next pass (or null);
// jzero: matrix to clear for the pass after next (or null); xf: bit 0 = map signed/float keys on
// load (first pass), bit 1 = map back on store (last pass)
template <int ES>
int launch_sweep(rsx_ctx* ctx, const void* src, void* dst, const RegionGeom& g, const rsx_layout* L, uint32_t digit,
                 const unsigned long long* J, unsigned long long* jnext, unsigned long long* jzero, int xf,
                 hipStream_t st);
// arrays of at most one tile (tile_elems(ES)): the whole sort in one launch of one workgroup, in place
template <int ES>
int launch_small_sort(rsx_ctx* ctx, void* data, size_t n, const rsx_layout* L, hipStream_t st);
template <int ES>
int launch_segcopy(rsx_ctx* ctx, const void* src, void* dst, const uint64_t* so, const uint64_t* dof,
                   const uint64_t* len, uint32_t nseg, hipStream_t st);

}  // namespace rsxh
