// rsx.hip -- host side of librsx.so: the C-ABI of include/rsx.h over the gfx950
// kernels in rsx_device.hpp.  Plays the role of the body of
// `<[T]>::radix_sort` (reference src/radix_sort/mod.rs:62-175): pass loop,
// ping-pong, odd-D copy-back -- with every phase a stream-ordered launch.
#include "rsx_device.hpp"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rsx.h"

using namespace rsx;

namespace {
// aux block layout (one hipMalloc, zeroed at creation)
constexpr size_t J_BYTES = (size_t)MAX_REGIONS * RADIX * sizeof(uint64_t);  // one count matrix
constexpr size_t OFF_J0 = 0;
constexpr size_t OFF_J1 = OFF_J0 + J_BYTES;
constexpr size_t OFF_BASE = OFF_J1 + J_BYTES;       // [MAX_REGIONS][256] write cursors
constexpr size_t OFF_TICKETS = OFF_BASE + J_BYTES;  // [MAX_REGIONS] u32
constexpr size_t OFF_ERROR = OFF_TICKETS + 192;  // tickets[MAX_REGIONS] + roll-call words
constexpr size_t OFF_DBG = OFF_ERROR + 64;  // 8 diagnostic counters
constexpr size_t AUX_BYTES = OFF_DBG + 1024;  // 16 waves x 8 diagnostic counters
}  // namespace

struct rsx_ctx {
    int device = 0;
    std::mutex mu;
    std::string err = "";
    void* status = nullptr;  // tile status words, zeroed before every pass
    size_t status_bytes = 0;
    char* aux = nullptr;
    // staging for rsx_sort_host
    void* host_buf[2] = {nullptr, nullptr};
    size_t host_bytes = 0;
    int num_cu = 256;
    uint32_t pass_index = 0;   // of the sweep being launched within its sort (selects the status half)
    bool pass_last = true;     // no pass follows: nothing to clean
    bool rank_atomic = false;  // LDS atomic ordering self-test passed (set when the workspace is first made)
    uint32_t hot_lanes = 16;   // RSX_HOT env (tuning)
    uint32_t dbg = 0;  // RSX_DEBUG env: timing-only ablation switches for the sweep kernel
    // per-launch HIP-event timing (rsx_ctx_profile)
    bool prof = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pending[RSX_PROF_KINDS];
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_free;
    double prof_ms[RSX_PROF_KINDS] = {0, 0, 0, 0};
    uint64_t prof_n[RSX_PROF_KINDS] = {0, 0, 0, 0};
};

namespace {

int fail(rsx_ctx* c, int code, const char* what, hipError_t e = hipSuccess) {
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

bool layout_ok(const rsx_layout* L) {
    if (!L) return false;
    const uint32_t kb = L->key_bytes;
    if (!(kb == 1 || kb == 2 || kb == 4 || kb == 8 || kb == 16)) return false;
    if (L->key_kind > RSX_KEY_FLOAT) return false;
    if (L->key_kind == RSX_KEY_FLOAT && !(kb == 4 || kb == 8)) return false;
    if (L->elem_bytes == 0 || (uint64_t)L->key_offset + kb > L->elem_bytes) return false;
    return true;
}
bool size_supported(uint32_t es) {
    return es == 1 || es == 2 || es == 4 || es == 8 || es == 12 || es == 16 || es == 24 || es == 32;
}
uint32_t elem_align(uint32_t es) {
    switch (es) {
        case 1: return 1;
        case 2: return 2;
        case 4: case 12: return 4;
        case 8: case 24: return 8;
        default: return 16;
    }
}
bool aligned(const void* p, uint32_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

// Keys per thread by element size.  Tiles need not be powers of two (the last tile of a region is
// partial anyway); bigger tiles mean longer output runs per digit (fewer partial cache lines, the
// memory system's real cost here) and fewer look-backs per key, as long as two or three workgroups
// still fit a CU: u32 28 x 512 = 14336 keys (56 KiB), u64 12 x 512 (48 KiB), 16-byte 5 x 512 (40 KiB),
// 12-byte 10 x 512 (60 KiB), 24/32-byte 3 x 512 (36/48 KiB).
// Measured against 16 / 8 / 4: 1B u32 117 -> 136, 1B u64 32 -> 34.9, 128M (u64,u64) 17.3 -> 18 Gkeys/s.
#ifndef RSX_KPT4
#define RSX_KPT4 28
#endif
#ifndef RSX_WG4
#define RSX_WG4 512
#endif
#ifndef RSX_KPT8
#define RSX_KPT8 12
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
#ifndef RSX_WG8
#define RSX_WG8 512
#endif
constexpr int kpt_for(int es) { return es <= 4 ? RSX_KPT4 : es == 8 ? RSX_KPT8 : es == 12 ? RSX_KPT12 : es == 16 ? RSX_KPT16 : RSX_KPT32; }
constexpr int wg_for(int es) { return es <= 4 ? RSX_WG4 : es == 8 ? RSX_WG8 : 512; }
constexpr uint32_t tile_elems(int es) { return wg_for(es) * kpt_for(es); }

uint32_t log2u(uint64_t x) { return 63u - (uint32_t)__builtin_clzll(x); }

// Regions: smallest power-of-two length (>= one tile) that covers n with <= MAX_REGIONS of them.
RegionGeom make_geom(uint64_t n, uint32_t es) {
    RegionGeom g;
    g.n = n;
    uint32_t k = log2u(tile_elems((int)es));
    if ((1ull << k) < tile_elems((int)es)) ++k;  // tiles need not be a power of two; regions are
    static const uint64_t max_regions = [] {  // RSX_REGIONS env: tuning/diagnostics only
        const char* e = std::getenv("RSX_REGIONS");
        const uint64_t v = e ? std::strtoull(e, nullptr, 0) : 0;
        return (v >= 1 && v <= (uint64_t)MAX_REGIONS) ? v : 0;
    }();
    // the next pass's count matrix costs 1 KiB of LDS per region: 8 where the tile needs the room
    const uint64_t cap = max_regions ? max_regions : (es == 8 || es > 16) ? 16 : 8;
    while (((n + (1ull << k) - 1) >> k) > cap) ++k;
    g.region_shift = k;
    g.num_regions = (uint32_t)((n + (1ull << k) - 1) >> k);
    if (g.num_regions == 0) g.num_regions = 1;
    return g;
}
uint64_t tiles_per_region(const RegionGeom& g, uint32_t es) {
    const uint64_t t = tile_elems((int)es);
    return ((1ull << g.region_shift) + t - 1) / t;
}
uint64_t status_rows(const RegionGeom& g, uint32_t es) {
    return (uint64_t)g.num_regions * tiles_per_region(g, es);
}
// chain prefixes are relative to the region: 30 value bits suffice up to 2^30-element regions
bool status32(const RegionGeom& g) { return g.region_shift <= 30; }

DigitSpec make_spec(const rsx_layout* L, uint32_t digit) {
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

unsigned long long* J_of(rsx_ctx* c, int which) {
    return reinterpret_cast<unsigned long long*>(c->aux + (which ? OFF_J1 : OFF_J0));
}
uint64_t* base_of(rsx_ctx* c) { return reinterpret_cast<uint64_t*>(c->aux + OFF_BASE); }
uint32_t* tickets_of(rsx_ctx* c) { return reinterpret_cast<uint32_t*>(c->aux + OFF_TICKETS); }
uint32_t* error_of(rsx_ctx* c) { return reinterpret_cast<uint32_t*>(c->aux + OFF_ERROR); }

size_t status_bytes_for(size_t n, uint32_t es) {
    const RegionGeom g = make_geom(n, es);
    return (size_t)status_rows(g, es) * RADIX * (status32(g) ? 4 : 8);
}

int ensure_workspace(rsx_ctx* ctx, size_t n, const rsx_layout* L) {
    if (!ctx->aux) {
        void* p = nullptr;
        RSX_HIP(hipMalloc(&p, AUX_BYTES));
        ctx->aux = static_cast<char*>(p);
        RSX_HIP(hipMemset(ctx->aux, 0, AUX_BYTES));
        // may the sweep rank by returned LDS atomics on this device?  (see rsx_lds_order_kernel)
        uint32_t* flag = error_of(ctx) + 1;
        hipLaunchKernelGGL(rsx_lds_order_kernel, dim3(64), dim3(512), 0, nullptr, flag);
        RSX_HIP(hipGetLastError());
        uint32_t failed = 1;
        RSX_HIP(hipMemcpy(&failed, flag, sizeof failed, hipMemcpyDeviceToHost));
        ctx->rank_atomic = failed == 0 && !(ctx->dbg & 0x10000u);
        if (ctx->dbg & 0x200u) std::fprintf(stderr, "[rsx] LDS atomic order self-test %s\n", failed ? "FAILED: ballots only" : "passed");
    }
    const size_t need = status_bytes_for(n, L->elem_bytes);
    if (need > ctx->status_bytes) {
        if (ctx->status) RSX_HIP(hipFree(ctx->status));
        ctx->status = nullptr;
        ctx->status_bytes = 0;
        hipError_t e = hipMalloc(&ctx->status, 2 * need);  // two arrays: this pass's and the next pass's
        if (e != hipSuccess) return fail(ctx, RSX_ERR_NOMEM, "workspace hipMalloc", e);
        ctx->status_bytes = need;
    }
    return RSX_OK;
}

// ---- count phase of a first pass: J[r][v] for `digit` over the input regions ------------------
template <int ES, bool FLT>
int launch_hist_t(rsx_ctx* ctx, const void* src, const RegionGeom& g, const rsx_layout* L, uint32_t digit,
                  unsigned long long* J, hipStream_t st) {
    RSX_HIP(hipMemsetAsync(J, 0, J_BYTES, st));
    const uint64_t per_block = 512ull * 16;
    uint64_t bpr = ((1ull << g.region_shift) + per_block - 1) / per_block;
    const uint64_t cap = ((uint64_t)ctx->num_cu * 8 + g.num_regions - 1) / g.num_regions;
    if (bpr > cap) bpr = cap;
    if (bpr == 0) bpr = 1;
    LaunchTimer lt(ctx, RSX_PROF_HIST, st);
    hipLaunchKernelGGL((rsx_hist_kernel<ES, FLT>), dim3((uint32_t)(bpr * g.num_regions)), dim3(512), 0, st,
                       static_cast<const Elem<ES>*>(src), g, make_spec(L, digit), (uint32_t)bpr, J);
    RSX_HIP(hipGetLastError());
    return RSX_OK;
}
template <int ES>
int launch_hist(rsx_ctx* ctx, const void* src, const RegionGeom& g, const rsx_layout* L, uint32_t digit,
                unsigned long long* J, hipStream_t st) {
    if (L->key_kind == RSX_KEY_FLOAT || (L->key_kind == RSX_KEY_SIGNED && digit + 1 == L->key_bytes))
        return launch_hist_t<ES, true>(ctx, src, g, L, digit, J, st);
    return launch_hist_t<ES, false>(ctx, src, g, L, digit, J, st);
}

// ---- prefix phase ------------------------------------------------------------------------------
int launch_prefix(rsx_ctx* ctx, const RegionGeom& g, const unsigned long long* J, unsigned long long* jnext,
                  uint64_t* counts_out, hipStream_t st) {
    LaunchTimer lt(ctx, RSX_PROF_SCAN, st);
    hipLaunchKernelGGL(rsx_prefix_kernel, dim3(1), dim3(RADIX), 0, st, J, g.num_regions, base_of(ctx), jnext,
                       tickets_of(ctx), counts_out);
    RSX_HIP(hipGetLastError());
    return RSX_OK;
}

// ---- scatter phase: one sweep pass -------------------------------------------------------------
// the per-dword masks of the signed/float key map (KeyXform in rsx_device.hpp)
KeyXform make_xform(const rsx_layout* L) {
    KeyXform x;
    std::memset(&x, 0, sizeof x);
    if (L->key_kind == RSX_KEY_UNSIGNED) return x;
    const uint32_t top = L->key_offset + L->key_bytes - 1;
    auto word_of = [&](uint32_t byte) { return L->elem_bytes >= 4 ? byte >> 2 : 0u; };
    auto bit_of = [&](uint32_t byte) { return L->elem_bytes >= 4 ? 8 * (byte & 3) : 8 * byte; };
    const uint32_t sw = word_of(top);
    const uint32_t sbit = 1u << (bit_of(top) + 7);
    x.sign[sw] = sbit;
    x.xpos[sw] = sbit;
    if (L->key_kind == RSX_KEY_SIGNED) {
        x.xneg[sw] = sbit;
    } else {
        for (uint32_t b = L->key_offset; b <= top; ++b) x.xneg[word_of(b)] |= 0xFFu << bit_of(b);
    }
    return x;
}

template <int ES, typename S, int XF, bool NEXT>
int launch_sweep_t(rsx_ctx* ctx, const void* src, void* dst, const RegionGeom& g, const rsx_layout* L,
                   uint32_t digit, unsigned long long* jnext, hipStream_t st) {
    constexpr int KPT = kpt_for(ES);
    constexpr int SWEEP_WG = wg_for(ES);
    constexpr int TILE = SWEEP_WG * KPT;
    const uint64_t rows = status_rows(g, ES);
    // Status words alternate between the two halves of the workspace.  Only the first pass of a
    // sort zeroes its half with a memset; every pass zeroes, tile by tile, the half of the next.
    char* const half[2] = {static_cast<char*>(ctx->status), static_cast<char*>(ctx->status) + ctx->status_bytes};
    const uint32_t which = ctx->pass_index & 1u;
    if (ctx->pass_index == 0) RSX_HIP(hipMemsetAsync(half[0], 0, (size_t)rows * RADIX * sizeof(S), st));
    SweepArgs a;
    a.status_clean = ctx->pass_last ? nullptr : half[which ^ 1u];
    a.src = src;
    a.dst = dst;
    a.g = g;
    a.region_base = base_of(ctx);
    a.status = half[which];
    a.tickets = tickets_of(ctx);
    a.jnext = jnext;
    a.error = error_of(ctx);
    a.spec = make_spec(L, digit);
    a.next = make_spec(L, NEXT ? digit + 1 : digit);
    a.spec.flip = a.next.flip = 0;  // the sweep sees mapped keys: plain digits
    a.xf = make_xform(L);
    a.tiles_per_region = (uint32_t)tiles_per_region(g, ES);
    a.dbg = ctx->dbg;
    a.rank_atomic = ctx->rank_atomic ? 1u : 0u;
    a.hot_lanes = (ctx->dbg & 0x20000u) ? 65u : ctx->hot_lanes;  // 0x20000: atomics whatever the skew (timing only)
    a.dbg_cnt = reinterpret_cast<unsigned long long*>(ctx->aux + OFF_DBG);
    const size_t lds = (size_t)TILE * ES + (SWEEP_WG / WAVE) * RADIX * ((RSX_WIDE_CNT && ES <= 4 && KPT >= 16 && SWEEP_WG <= 512) ? sizeof(uint32_t) : sizeof(uint16_t)) +
                       (NEXT ? (size_t)g.num_regions * RADIX * sizeof(uint32_t) : 0) + 64;
    auto kern = rsx_sweep_kernel<ES, KPT, SWEEP_WG, S, XF, NEXT>;
    // resident workgroups per CU for this kernel at this LDS size (the count matrix of the next pass
    // makes the LDS size depend on the number of regions): cached per instantiation and thread
    thread_local size_t occ_lds = ~(size_t)0;
    thread_local int occ = 0;
    if (occ == 0 || occ_lds != lds) {
        int o = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, kern, SWEEP_WG, lds) != hipSuccess || o < 1) o = 2;
        occ = o;
        occ_lds = lds;
    }
    // persistent workgroups; correctness does not need them co-resident (a workgroup only
    // ever waits for tiles whose tickets were drawn earlier, by workgroups already running)
    const uint64_t total_tiles = (g.n + TILE - 1) / TILE + g.num_regions;
    uint64_t grid = (uint64_t)ctx->num_cu * occ;
    if (grid > total_tiles) grid = total_tiles;
    if (const char* o = std::getenv("RSX_OCC")) grid = (uint64_t)ctx->num_cu * std::atoi(o);  // tuning only
    {   // static mode: workgroups per region, proportional to the region's tile count, >= 1 each
        const uint32_t NR = g.num_regions;
        const uint64_t tpr = tiles_per_region(g, ES);
        const uint64_t real_tiles = (g.n + TILE - 1) / TILE;
        uint64_t cum = 0;
        for (uint32_t r = 0; r < NR; ++r) {
            a.wg_first[r] = (uint16_t)(cum * grid / real_tiles);
            const uint64_t left = real_tiles - cum;
            cum += left < tpr ? left : tpr;
        }
        a.wg_first[NR] = (uint16_t)grid;
        for (uint32_t r = 0; r < NR; ++r)  // at least one workgroup per region
            if (a.wg_first[r + 1] <= a.wg_first[r]) a.wg_first[r + 1] = a.wg_first[r] + 1;
        for (uint32_t r = NR; r-- > 0;) {
            const uint32_t cap = (uint32_t)grid - (NR - r);
            if (a.wg_first[r] > cap) a.wg_first[r] = (uint16_t)cap;
        }
        a.wg_first[NR] = (uint16_t)grid;
        for (uint32_t r = NR + 1; r <= (uint32_t)MAX_REGIONS; ++r) a.wg_first[r] = (uint16_t)grid;
        // regions whose workgroups fall into one class of the kernel's XCD-major numbering
        // (class = index / (grid/8) = blockIdx % 8): candidates for L2-local status words
        a.local_mask = 0;
        if (grid % 8 == 0 && !(ctx->dbg & 0x4000u))
            for (uint32_t r = 0; r < NR; ++r)
                if (a.wg_first[r] / (grid / 8) == (a.wg_first[r + 1] - 1u) / (grid / 8)) a.local_mask |= 1u << r;
    }
    if (ctx->dbg & 0x200u) std::fprintf(stderr, "[rsx] sweep ES=%d NEXT=%d occ=%d grid=%llu lds=%zu tiles=%llu regions=%u\n", ES, (int)NEXT, occ, (unsigned long long)grid, lds, (unsigned long long)total_tiles, g.num_regions);
    LaunchTimer lt(ctx, RSX_PROF_SWEEP, st);
    hipLaunchKernelGGL(kern, dim3((uint32_t)grid), dim3(SWEEP_WG), lds, st, a);
    RSX_HIP(hipGetLastError());
    return RSX_OK;
}

template <int ES, typename S, int XF>
int launch_sweep_n(rsx_ctx* ctx, const void* src, void* dst, const RegionGeom& g, const rsx_layout* L,
                   uint32_t digit, unsigned long long* jnext, hipStream_t st) {
    if (jnext) return launch_sweep_t<ES, S, XF, true>(ctx, src, dst, g, L, digit, jnext, st);
    return launch_sweep_t<ES, S, XF, false>(ctx, src, dst, g, L, digit, jnext, st);
}

// xf: bit 0 = map signed/float keys on load (first pass), bit 1 = map back on store (last pass)
template <int ES, typename S>
int launch_sweep_x(rsx_ctx* ctx, const void* src, void* dst, const RegionGeom& g, const rsx_layout* L,
                   uint32_t digit, unsigned long long* jnext, int xf, hipStream_t st) {
    switch (L->key_kind == RSX_KEY_UNSIGNED ? 0 : xf) {
        case 1: return launch_sweep_n<ES, S, 1>(ctx, src, dst, g, L, digit, jnext, st);
        case 2: return launch_sweep_n<ES, S, 2>(ctx, src, dst, g, L, digit, jnext, st);
        case 3: return launch_sweep_n<ES, S, 3>(ctx, src, dst, g, L, digit, jnext, st);
        default: return launch_sweep_n<ES, S, 0>(ctx, src, dst, g, L, digit, jnext, st);
    }
}

template <int ES>
int launch_sweep(rsx_ctx* ctx, const void* src, void* dst, const RegionGeom& g, const rsx_layout* L, uint32_t digit,
                 unsigned long long* jnext, int xf, hipStream_t st) {
    if (status32(g)) return launch_sweep_x<ES, uint32_t>(ctx, src, dst, g, L, digit, jnext, xf, st);
    return launch_sweep_x<ES, uint64_t>(ctx, src, dst, g, L, digit, jnext, xf, st);
}

#define RSX_DISPATCH_ES(es, FN, ...)                           \
    switch (es) {                                              \
        case 1: return FN<1>(__VA_ARGS__);                     \
        case 2: return FN<2>(__VA_ARGS__);                     \
        case 4: return FN<4>(__VA_ARGS__);                     \
        case 8: return FN<8>(__VA_ARGS__);                     \
        case 12: return FN<12>(__VA_ARGS__);                   \
        case 16: return FN<16>(__VA_ARGS__);                   \
        case 24: return FN<24>(__VA_ARGS__);                   \
        case 32: return FN<32>(__VA_ARGS__);                   \
        default: return fail(ctx, RSX_ERR_UNSUPPORTED, "element size has no device kernel"); \
    }

int hist_dispatch(rsx_ctx* ctx, const void* src, const RegionGeom& g, const rsx_layout* L, uint32_t digit,
                  unsigned long long* J, hipStream_t st) {
    RSX_DISPATCH_ES(L->elem_bytes, launch_hist, ctx, src, g, L, digit, J, st)
}
int sweep_dispatch(rsx_ctx* ctx, const void* src, void* dst, const RegionGeom& g, const rsx_layout* L,
                   uint32_t digit, unsigned long long* jnext, int xf, hipStream_t st) {
    RSX_DISPATCH_ES(L->elem_bytes, launch_sweep, ctx, src, dst, g, L, digit, jnext, xf, st)
}

template <int ES>
int launch_segcopy(rsx_ctx* ctx, const void* src, void* dst, const uint64_t* so, const uint64_t* dof,
                   const uint64_t* len, uint32_t nseg, hipStream_t st) {
    const uint32_t bps = 8;
    hipLaunchKernelGGL((rsx_segcopy_kernel<ES>), dim3(nseg * bps), dim3(256), 0, st,
                       static_cast<const Elem<ES>*>(src), static_cast<Elem<ES>*>(dst), so, dof, len, nseg, bps);
    RSX_HIP(hipGetLastError());
    return RSX_OK;
}

int check_common(rsx_ctx* ctx, const rsx_layout* L) {
    if (!ctx) return RSX_ERR_ARG;
    if (!layout_ok(L)) return fail(ctx, RSX_ERR_ARG, "invalid rsx_layout");
    if (!size_supported(L->elem_bytes)) return fail(ctx, RSX_ERR_UNSUPPORTED, "element size has no device kernel");
    return RSX_OK;
}

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace

extern "C" {

int rsx_version(void) { return RSX_VERSION; }

const char* rsx_strerror(int status) {
    switch (status) {
        case RSX_OK: return "ok";
        case RSX_ERR_ARG: return "invalid argument";
        case RSX_ERR_UNSUPPORTED: return "unsupported element layout";
        case RSX_ERR_HIP: return "HIP runtime error";
        case RSX_ERR_NOMEM: return "out of device memory";
        case RSX_ERR_NODEVICE: return "no usable device";
        case RSX_ERR_WORKSPACE: return "workspace not reserved";
        case RSX_ERR_INTERNAL: return "device-side protocol error";
        default: return "unknown status";
    }
}

const char* rsx_last_error(const rsx_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int rsx_ctx_create(int device, rsx_ctx** out) try {
    if (!out) return RSX_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return RSX_ERR_NODEVICE;
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) return RSX_ERR_NODEVICE;
    }
    if (device >= count) return RSX_ERR_NODEVICE;
    rsx_ctx* ctx = new (std::nothrow) rsx_ctx();
    if (!ctx) return RSX_ERR_NOMEM;
    ctx->device = device;
    if (const char* dbg = std::getenv("RSX_DEBUG")) ctx->dbg = (uint32_t)std::strtoul(dbg, nullptr, 0);
    if (const char* h = std::getenv("RSX_HOT")) ctx->hot_lanes = (uint32_t)std::strtoul(h, nullptr, 0);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {  // code objects are gfx950-only
            delete ctx;
            return RSX_ERR_NODEVICE;
        }
    }
    *out = ctx;
    return RSX_OK;
} catch (...) {
    return RSX_ERR_NOMEM;
}

int rsx_ctx_destroy(rsx_ctx* ctx) try {
    if (!ctx) return RSX_ERR_ARG;
    {
        DeviceGuard g(ctx->device);
        if (ctx->status) (void)hipFree(ctx->status);
        if (ctx->aux) (void)hipFree(ctx->aux);
        for (void* p : ctx->host_buf)
            if (p) (void)hipFree(p);
        for (int k = 0; k < RSX_PROF_KINDS; ++k)
            for (auto& e : ctx->prof_pending[k]) ctx->prof_free.push_back(e);
        for (auto& e : ctx->prof_free) {
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
    }
    delete ctx;
    return RSX_OK;
} catch (...) {
    return RSX_ERR_HIP;
}

int rsx_ctx_reserve(rsx_ctx* ctx, size_t n, const rsx_layout* layout) try {
    int rc = check_common(ctx, layout);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    return ensure_workspace(ctx, n, layout);
} catch (...) {
    return RSX_ERR_NOMEM;
}

int rsx_ctx_check(rsx_ctx* ctx, void* stream) try {
    if (!ctx) return RSX_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    RSX_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    if (!ctx->aux) return RSX_OK;
    uint32_t e = 0;
    RSX_HIP(hipMemcpy(&e, error_of(ctx), sizeof e, hipMemcpyDeviceToHost));
    if (e) {
        (void)hipMemset(error_of(ctx), 0, sizeof e);
        return fail(ctx, RSX_ERR_INTERNAL, "look-back spin gave up (device protocol error)");
    }
    return RSX_OK;
} catch (...) {
    return RSX_ERR_HIP;
}

// Diagnostic counters of the sweep kernel (RSX_DEBUG & 0x100); not part of include/rsx.h.
int rsx_debug_counters(rsx_ctx* ctx, unsigned long long* out8, int reset) try {
    if (!ctx || !out8 || !ctx->aux) return RSX_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    RSX_HIP(hipDeviceSynchronize());
    RSX_HIP(hipMemcpy(out8, ctx->aux + OFF_DBG, 1024, hipMemcpyDeviceToHost));  // caller passes 128 u64
    if (reset) RSX_HIP(hipMemset(ctx->aux + OFF_DBG, 0, 1024));
    return RSX_OK;
} catch (...) {
    return RSX_ERR_HIP;
}

int rsx_ctx_profile(rsx_ctx* ctx, int enable) try {
    if (!ctx) return RSX_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (enable) {
        for (int k = 0; k < RSX_PROF_KINDS; ++k) {
            for (auto& e : ctx->prof_pending[k]) ctx->prof_free.push_back(e);
            ctx->prof_pending[k].clear();
            ctx->prof_ms[k] = 0;
            ctx->prof_n[k] = 0;
        }
    }
    ctx->prof = enable != 0;
    return RSX_OK;
} catch (...) {
    return RSX_ERR_NOMEM;
}

int rsx_ctx_profile_read(rsx_ctx* ctx, double* ms, uint64_t* launches) try {
    if (!ctx || !ms || !launches) return RSX_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    for (int k = 0; k < RSX_PROF_KINDS; ++k) {
        for (auto& e : ctx->prof_pending[k]) {
            RSX_HIP(hipEventSynchronize(e.second));
            float t = 0;
            RSX_HIP(hipEventElapsedTime(&t, e.first, e.second));
            ctx->prof_ms[k] += t;
            ctx->prof_n[k] += 1;
            ctx->prof_free.push_back(e);
        }
        ctx->prof_pending[k].clear();
        ms[k] = ctx->prof_ms[k];
        launches[k] = ctx->prof_n[k];
    }
    return RSX_OK;
} catch (...) {
    return RSX_ERR_HIP;
}

int rsx_sort_device(rsx_ctx* ctx, void* d_data, void* d_tmp, size_t n, const rsx_layout* L, void* stream) try {
    int rc = check_common(ctx, L);
    if (rc) return rc;
    if (n <= 1) return RSX_OK;  // reference panics on n == 0 (mod.rs:66-70,92); nothing to compare
    if (!d_data || !d_tmp) return fail(ctx, RSX_ERR_ARG, "null device pointer");
    const uint32_t al = elem_align(L->elem_bytes);
    if (!aligned(d_data, al) || !aligned(d_tmp, al)) return fail(ctx, RSX_ERR_ARG, "device pointer misaligned");
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    if (!g.ok) return fail(ctx, RSX_ERR_NODEVICE, "hipSetDevice failed");
    rc = ensure_workspace(ctx, n, L);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const uint32_t D = L->key_bytes;  // T::NUMBER_OF_DIGITS
    const RegionGeom geom = make_geom(n, L->elem_bytes);
    // count phase of pass 0 (mod.rs:90-109); later passes are counted by the sweep before them
    rc = hist_dispatch(ctx, d_data, geom, L, 0, J_of(ctx, 0), st);
    if (rc) return rc;
    if (L->elem_bytes == 1 && !(ctx->dbg & 0x40000u)) {
        // u8 / i8: the element is its digit, so the 256 counts ARE the sorted array (same bytes as
        // the pass + copy-back of mod.rs:121-174 would leave): write the runs, skip scatter and copy
        uint64_t* totals = reinterpret_cast<uint64_t*>(J_of(ctx, 1));
        rc = launch_prefix(ctx, geom, J_of(ctx, 0), nullptr, totals, st);
        if (rc) return rc;
        LaunchTimer lt(ctx, RSX_PROF_OTHER, st);
        const uint64_t chunks = (n + 15) / 16;
        uint64_t blocks = (chunks + 255) / 256;
        if (blocks > (uint64_t)ctx->num_cu * 16) blocks = (uint64_t)ctx->num_cu * 16;
        hipLaunchKernelGGL(rsx_expand_bytes_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, static_cast<uint8_t*>(d_data),
                           (uint64_t)n, totals, L->key_kind == RSX_KEY_SIGNED ? 0x80u : 0u);
        RSX_HIP(hipGetLastError());
        return RSX_OK;
    }
    // pass loop with ping-pong (mod.rs:84-89)
    for (uint32_t d = 0; d < D; ++d) {
        const void* src = (d % 2 == 0) ? d_data : d_tmp;
        void* dst = (d % 2 == 0) ? d_tmp : d_data;
        unsigned long long* jnext = (d + 1 < D) ? J_of(ctx, (d + 1) & 1) : nullptr;
        rc = launch_prefix(ctx, geom, J_of(ctx, d & 1), jnext, nullptr, st);  // mod.rs:110-120
        if (rc) return rc;
        const int xf = (d == 0 ? 1 : 0) | (d + 1 == D ? 2 : 0);  // key map on at the first, off at the last pass
        ctx->pass_index = d;
        ctx->pass_last = d + 1 == D;
        rc = sweep_dispatch(ctx, src, dst, geom, L, d, jnext, xf, st);  // mod.rs:121-168
        if (rc) return rc;
    }
    if (D % 2 == 1)  // odd-D copy-back (mod.rs:170-174)
        RSX_HIP(hipMemcpyAsync(d_data, d_tmp, n * (size_t)L->elem_bytes, hipMemcpyDeviceToDevice, st));
    return RSX_OK;
} catch (...) {
    return RSX_ERR_HIP;
}

int rsx_sort_host(rsx_ctx* ctx, void* data, size_t n, const rsx_layout* L) try {
    int rc = check_common(ctx, L);
    if (rc) return rc;
    if (n <= 1) return RSX_OK;
    if (!data) return fail(ctx, RSX_ERR_ARG, "null host pointer");
    const size_t bytes = n * (size_t)L->elem_bytes;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        DeviceGuard g(ctx->device);
        if (bytes > ctx->host_bytes) {
            for (void*& p : ctx->host_buf) {
                if (p) (void)hipFree(p);
                p = nullptr;
            }
            ctx->host_bytes = 0;
            for (void*& p : ctx->host_buf) {
                hipError_t e = hipMalloc(&p, bytes);
                if (e != hipSuccess) return fail(ctx, RSX_ERR_NOMEM, "staging hipMalloc", e);
            }
            ctx->host_bytes = bytes;
        }
        RSX_HIP(hipMemcpy(ctx->host_buf[0], data, bytes, hipMemcpyHostToDevice));
    }
    rc = rsx_sort_device(ctx, ctx->host_buf[0], ctx->host_buf[1], n, L, nullptr);
    if (rc) return rc;
    rc = rsx_ctx_check(ctx, nullptr);
    if (rc) return rc;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        DeviceGuard g(ctx->device);
        RSX_HIP(hipMemcpy(data, ctx->host_buf[0], bytes, hipMemcpyDeviceToHost));
    }
    return RSX_OK;
} catch (...) {
    return RSX_ERR_HIP;
}

int rsx_histogram_device(rsx_ctx* ctx, const void* d_src, size_t n, const rsx_layout* L, uint32_t digit,
                         uint64_t* d_hist, void* stream) try {
    int rc = check_common(ctx, L);
    if (rc) return rc;
    if (digit >= L->key_bytes || !d_hist) return fail(ctx, RSX_ERR_ARG, "bad digit / null histogram");
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n == 0) {
        RSX_HIP(hipMemsetAsync(d_hist, 0, RADIX * sizeof(uint64_t), st));
        return RSX_OK;
    }
    if (!d_src || !aligned(d_src, elem_align(L->elem_bytes))) return fail(ctx, RSX_ERR_ARG, "bad source pointer");
    rc = ensure_workspace(ctx, n, L);
    if (rc) return rc;
    const RegionGeom geom = make_geom(n, L->elem_bytes);
    rc = hist_dispatch(ctx, d_src, geom, L, digit, J_of(ctx, 0), st);
    if (rc) return rc;
    return launch_prefix(ctx, geom, J_of(ctx, 0), nullptr, d_hist, st);  // column sums -> d_hist
} catch (...) {
    return RSX_ERR_HIP;
}

int rsx_partition_device(rsx_ctx* ctx, const void* d_src, void* d_dst, size_t n, const rsx_layout* L,
                         uint32_t digit, uint64_t* d_hist, void* stream) try {
    int rc = check_common(ctx, L);
    if (rc) return rc;
    if (digit >= L->key_bytes) return fail(ctx, RSX_ERR_ARG, "bad digit");
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n == 0) {
        if (d_hist) RSX_HIP(hipMemsetAsync(d_hist, 0, RADIX * sizeof(uint64_t), st));
        return RSX_OK;
    }
    const uint32_t al = elem_align(L->elem_bytes);
    if (!d_src || !d_dst || !aligned(d_src, al) || !aligned(d_dst, al))
        return fail(ctx, RSX_ERR_ARG, "bad device pointer");
    rc = ensure_workspace(ctx, n, L);
    if (rc) return rc;
    const RegionGeom geom = make_geom(n, L->elem_bytes);
    rc = hist_dispatch(ctx, d_src, geom, L, digit, J_of(ctx, 0), st);
    if (rc) return rc;
    rc = launch_prefix(ctx, geom, J_of(ctx, 0), nullptr, d_hist, st);
    if (rc) return rc;
    ctx->pass_index = 0;
    ctx->pass_last = true;
    return sweep_dispatch(ctx, d_src, d_dst, geom, L, digit, nullptr, 3, st);  // a lone pass maps and unmaps
} catch (...) {
    return RSX_ERR_HIP;
}

int rsx_segmented_copy_device(rsx_ctx* ctx, const void* d_src, void* d_dst, uint32_t elem_bytes,
                              const uint64_t* d_src_off, const uint64_t* d_dst_off, const uint64_t* d_len,
                              uint32_t nseg, void* stream) try {
    if (!ctx) return RSX_ERR_ARG;
    if (nseg == 0) return RSX_OK;
    if (!d_src || !d_dst || !d_src_off || !d_dst_off || !d_len) return fail(ctx, RSX_ERR_ARG, "null pointer");
    if (!size_supported(elem_bytes)) return fail(ctx, RSX_ERR_UNSUPPORTED, "element size has no device kernel");
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    RSX_DISPATCH_ES(elem_bytes, launch_segcopy, ctx, d_src, d_dst, d_src_off, d_dst_off, d_len, nseg, st)
} catch (...) {
    return RSX_ERR_HIP;
}

int rsx_extract_keys_device(rsx_ctx* ctx, const void* d_src, size_t n, const rsx_layout* L, int64_t* d_keys,
                            void* stream) try {
    if (!ctx) return RSX_ERR_ARG;
    if (!layout_ok(L)) return fail(ctx, RSX_ERR_ARG, "invalid rsx_layout");
    if (L->key_bytes > 8) return fail(ctx, RSX_ERR_UNSUPPORTED, "keys wider than 8 bytes have no 64-bit form");
    if (n == 0) return RSX_OK;
    if (!d_src || !d_keys) return fail(ctx, RSX_ERR_ARG, "null pointer");
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint64_t blocks = (n + 255) / 256;
    const uint64_t cap = (uint64_t)ctx->num_cu * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(rsx_extract_keys_kernel, dim3((uint32_t)blocks), dim3(256), 0, st,
                       static_cast<const uint8_t*>(d_src), (uint64_t)n, L->elem_bytes, L->key_offset, L->key_bytes,
                       L->key_kind, reinterpret_cast<long long*>(d_keys));
    RSX_HIP(hipGetLastError());
    return RSX_OK;
} catch (...) {
    return RSX_ERR_HIP;
}

int rsx_bounds_device(rsx_ctx* ctx, const void* d_sorted, size_t n, const rsx_layout* L, const uint64_t* d_queries,
                      uint32_t nq, uint64_t* d_out, void* stream) try {
    if (!ctx) return RSX_ERR_ARG;
    if (!layout_ok(L)) return fail(ctx, RSX_ERR_ARG, "invalid rsx_layout");
    if (nq == 0) return RSX_OK;
    if (!d_queries || !d_out || (n && !d_sorted)) return fail(ctx, RSX_ERR_ARG, "null pointer");
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(rsx_bounds_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, static_cast<const uint8_t*>(d_sorted),
                       (uint64_t)n, L->elem_bytes, L->key_offset, L->key_bytes, L->key_kind, d_queries, nq, d_out);
    RSX_HIP(hipGetLastError());
    return RSX_OK;
} catch (...) {
    return RSX_ERR_HIP;
}

// ---- multi-GPU, one process: one exchange between two local sorts ------------------------------
// The G slices are "chunks" in the sense of mod.rs:66-70; the result is what the reference would
// produce on their concatenation.  Schedule: (1) every device sorts its slice; (2) the G-1 slice
// boundaries of the sorted whole are located exactly -- boundary h is the key K_h with
// less(K_h) <= T_h < less_or_equal(K_h) over all slices, found digit by digit (256 candidates per
// step, counted by binary search in every sorted slice), ties on K_h dealt out in slice order;
// (3) each device pushes the G ranges of its slice to their owners over xGMI, ordered by source
// slice at the receiver; (4) a second stable local sort merges the G sorted runs.  Stability: equal
// keys stay in (source slice, local index) order through (3), and (4) is stable.
int rsx_sort_sharded(rsx_ctx* const* ctxs, uint32_t ndev, void* const* d_slices, void* const* d_tmps,
                     const size_t* n_per_dev, const rsx_layout* L) try {
    if (!ctxs || ndev == 0 || !ctxs[0]) return RSX_ERR_ARG;
    rsx_ctx* ctx = ctxs[0];  // carries the error text
    if (!d_slices || !d_tmps || !n_per_dev) return fail(ctx, RSX_ERR_ARG, "null table");
    const uint32_t G = ndev;
    if (G > 64) return fail(ctx, RSX_ERR_ARG, "more than 64 slices");
    for (uint32_t g = 0; g < G; ++g) {
        if (!ctxs[g]) return fail(ctx, RSX_ERR_ARG, "null context in table");
        int rc = check_common(ctxs[g], L);
        if (rc) return rc == RSX_ERR_ARG ? fail(ctx, rc, "invalid rsx_layout") : fail(ctx, rc, "element size has no device kernel");
        if (n_per_dev[g] && (!d_slices[g] || !d_tmps[g])) return fail(ctx, RSX_ERR_ARG, "null device pointer");
        for (uint32_t h = 0; h < g; ++h)
            if (ctxs[h] == ctxs[g]) return fail(ctx, RSX_ERR_ARG, "one context per slice");
    }
    const size_t es = L->elem_bytes;
    // (1) local sorts, all devices at once
    auto sort_all = [&]() -> int {
        for (uint32_t g = 0; g < G; ++g) {
            int rc = rsx_sort_device(ctxs[g], d_slices[g], d_tmps[g], n_per_dev[g], L, nullptr);
            if (rc) return g ? fail(ctx, rc, rsx_last_error(ctxs[g])) : rc;
        }
        for (uint32_t g = 0; g < G; ++g) {
            int rc = rsx_ctx_check(ctxs[g], nullptr);
            if (rc) return g ? fail(ctx, rc, rsx_last_error(ctxs[g])) : rc;
        }
        return RSX_OK;
    };
    int rc = sort_all();
    if (rc || G == 1) return rc;

    std::vector<uint64_t> bounds(G + 1, 0);
    for (uint32_t g = 0; g < G; ++g) bounds[g + 1] = bounds[g] + n_per_dev[g];
    const uint32_t nb = G - 1;
    const uint32_t nq_max = nb * RADIX;

    // per-device query / answer buffers
    struct Scratch {
        rsx_ctx* c = nullptr;
        uint64_t* q = nullptr;
        uint64_t* out = nullptr;
        ~Scratch() {
            if (!c) return;
            DeviceGuard g(c->device);
            if (q) (void)hipFree(q);
            if (out) (void)hipFree(out);
        }
    };
    std::vector<Scratch> scr(G);
    for (uint32_t g = 0; g < G; ++g) {
        DeviceGuard dg(ctxs[g]->device);
        if (!dg.ok) return fail(ctx, RSX_ERR_NODEVICE, "hipSetDevice failed");
        scr[g].c = ctxs[g];
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&scr[g].q), (size_t)nq_max * 2 * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&scr[g].out), (size_t)nq_max * 2 * sizeof(uint64_t));
        if (e != hipSuccess) return fail(ctx, RSX_ERR_NOMEM, "splitter scratch hipMalloc", e);
    }
    // counts[g][0..nq) = elements < Q on slice g, counts[g][nq..2nq) = elements <= Q
    std::vector<std::vector<uint64_t>> counts(G, std::vector<uint64_t>((size_t)nq_max * 2));
    auto ask = [&](const std::vector<uint64_t>& q, uint32_t nq) -> int {
        for (uint32_t g = 0; g < G; ++g) {
            std::lock_guard<std::mutex> lk(ctxs[g]->mu);
            DeviceGuard dg(ctxs[g]->device);
            RSX_HIP(hipMemcpyAsync(scr[g].q, q.data(), (size_t)nq * 2 * sizeof(uint64_t), hipMemcpyHostToDevice, nullptr));
            hipLaunchKernelGGL(rsx_bounds_kernel, dim3((nq + 255) / 256), dim3(256), 0, nullptr,
                               static_cast<const uint8_t*>(d_slices[g]), (uint64_t)n_per_dev[g], L->elem_bytes,
                               L->key_offset, L->key_bytes, L->key_kind, scr[g].q, nq, scr[g].out);
            RSX_HIP(hipGetLastError());
        }
        for (uint32_t g = 0; g < G; ++g) {
            std::lock_guard<std::mutex> lk(ctxs[g]->mu);
            DeviceGuard dg(ctxs[g]->device);
            RSX_HIP(hipMemcpy(counts[g].data(), scr[g].out, (size_t)nq * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost));
        }
        return RSX_OK;
    };

    // (2) splitters, most significant digit first
    std::vector<uint64_t> pre_lo(nb, 0), pre_hi(nb, 0), q((size_t)nq_max * 2);
    for (int digit = (int)L->key_bytes - 1; digit >= 0; --digit) {
        for (uint32_t b = 0; b < nb; ++b)
            for (uint32_t j = 0; j < RADIX; ++j) {
                uint64_t lo = pre_lo[b], hi = pre_hi[b];
                if (digit < 8) lo |= (uint64_t)j << (8 * digit);
                else hi |= (uint64_t)j << (8 * (digit - 8));
                q[2 * ((size_t)b * RADIX + j)] = lo;
                q[2 * ((size_t)b * RADIX + j) + 1] = hi;
            }
        rc = ask(q, nq_max);
        if (rc) return rc;
        for (uint32_t b = 0; b < nb; ++b) {
            // largest candidate whose global "less" count does not exceed the boundary
            uint32_t pick = 0;
            for (uint32_t j = 0; j < RADIX; ++j) {
                uint64_t less = 0;
                for (uint32_t g = 0; g < G; ++g) less += counts[g][(size_t)b * RADIX + j];
                if (less <= bounds[b + 1]) pick = j;  // monotone in j
            }
            if (digit < 8) pre_lo[b] |= (uint64_t)pick << (8 * digit);
            else pre_hi[b] |= (uint64_t)pick << (8 * (digit - 8));
        }
    }
    for (uint32_t b = 0; b < nb; ++b) {
        q[2 * b] = pre_lo[b];
        q[2 * b + 1] = pre_hi[b];
    }
    rc = ask(q, nb);
    if (rc) return rc;
    // split[g][h] = first element of slice g that goes to owner h
    std::vector<std::vector<uint64_t>> split(G, std::vector<uint64_t>(G + 1, 0));
    for (uint32_t g = 0; g < G; ++g) split[g][G] = n_per_dev[g];
    for (uint32_t b = 0; b < nb; ++b) {
        uint64_t less_total = 0;
        for (uint32_t g = 0; g < G; ++g) less_total += counts[g][b];
        uint64_t need = bounds[b + 1] - less_total;  // elements equal to K_b that go below the boundary
        for (uint32_t g = 0; g < G; ++g) {            // ties: lower slice first (stability)
            const uint64_t less = counts[g][b], eq = counts[g][nb + b] - less;
            const uint64_t take = need < eq ? need : eq;
            split[g][b + 1] = less + take;
            need -= take;
        }
        if (need != 0) return fail(ctx, RSX_ERR_INTERNAL, "splitter search inconsistent");
    }
    for (uint32_t h = 0; h < G; ++h) {
        uint64_t got = 0;
        for (uint32_t g = 0; g < G; ++g) {
            if (split[g][h + 1] < split[g][h]) return fail(ctx, RSX_ERR_INTERNAL, "splitters not monotone");
            got += split[g][h + 1] - split[g][h];
        }
        if (got != n_per_dev[h]) return fail(ctx, RSX_ERR_INTERNAL, "exchange plan does not fill a slice");
    }

    // (3) the exchange: slice g pushes its range for owner h into h's scratch, behind the ranges
    // of the slices before it.  d_tmps is idle (all local sorts were synchronised above).
    for (uint32_t g = 0; g < G; ++g) {
        std::lock_guard<std::mutex> lk(ctxs[g]->mu);
        DeviceGuard dg(ctxs[g]->device);
        for (uint32_t k = 0; k < G; ++k) {
            const uint32_t h = (g + k) % G;  // start with myself, then round the ring: spreads the links
            const uint64_t cnt = split[g][h + 1] - split[g][h];
            if (cnt == 0) continue;
            uint64_t at = 0;
            for (uint32_t p = 0; p < g; ++p) at += split[p][h + 1] - split[p][h];
            const char* src = static_cast<const char*>(d_slices[g]) + split[g][h] * es;
            char* dst = static_cast<char*>(d_tmps[h]) + at * es;
            if (ctxs[h]->device == ctxs[g]->device) {
                RSX_HIP(hipMemcpyAsync(dst, src, cnt * es, hipMemcpyDeviceToDevice, nullptr));
            } else {
                // direct xGMI writes where the topology allows; the copy works either way
                if (hipDeviceEnablePeerAccess(ctxs[h]->device, 0) != hipSuccess) (void)hipGetLastError();
                RSX_HIP(hipMemcpyPeerAsync(dst, ctxs[h]->device, src, ctxs[g]->device, cnt * es, nullptr));
            }
        }
    }
    for (uint32_t g = 0; g < G; ++g) {
        DeviceGuard dg(ctxs[g]->device);
        RSX_HIP(hipStreamSynchronize(nullptr));
    }
    // (4) G sorted runs per slice -> one: a stable sort of the received slice
    for (uint32_t g = 0; g < G; ++g) {
        if (n_per_dev[g] == 0) continue;
        DeviceGuard dg(ctxs[g]->device);
        RSX_HIP(hipMemcpyAsync(d_slices[g], d_tmps[g], n_per_dev[g] * es, hipMemcpyDeviceToDevice, nullptr));
    }
    return sort_all();
} catch (...) {
    return RSX_ERR_NOMEM;
}

int rsx_generate_device(rsx_ctx* ctx, void* d_data, size_t n, const rsx_layout* L, int gen, uint64_t seed,
                        double param, uint64_t index_base, void* stream) try {
    if (!ctx) return RSX_ERR_ARG;
    if (!layout_ok(L)) return fail(ctx, RSX_ERR_ARG, "invalid rsx_layout");
    if (n == 0) return RSX_OK;
    if (!d_data) return fail(ctx, RSX_ERR_ARG, "null pointer");
    if (gen < RSX_GEN_UNIFORM || gen > RSX_GEN_CONSTANT) return fail(ctx, RSX_ERR_ARG, "unknown generator");
    if (gen == RSX_GEN_STEP && !(param >= 1.0)) return fail(ctx, RSX_ERR_ARG, "step generator needs param >= 1");
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint64_t blocks = (n + 255) / 256;
    const uint64_t cap = (uint64_t)ctx->num_cu * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(rsx_generate_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, static_cast<uint8_t*>(d_data),
                       (uint64_t)n, L->elem_bytes, L->key_offset, L->key_bytes, gen, seed, param, index_base);
    RSX_HIP(hipGetLastError());
    return RSX_OK;
} catch (...) {
    return RSX_ERR_HIP;
}

int rsx_verify_device(rsx_ctx* ctx, const void* d_data, size_t n, const rsx_layout* L, uint64_t* d_out,
                      void* stream) try {
    if (!ctx) return RSX_ERR_ARG;
    if (!layout_ok(L)) return fail(ctx, RSX_ERR_ARG, "invalid rsx_layout");
    if (!d_out) return fail(ctx, RSX_ERR_ARG, "null pointer");
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    RSX_HIP(hipMemsetAsync(d_out, 0, 3 * sizeof(uint64_t), st));
    if (n == 0) return RSX_OK;
    if (!d_data) return fail(ctx, RSX_ERR_ARG, "null pointer");
    uint64_t blocks = (n + 255) / 256;
    const uint64_t cap = (uint64_t)ctx->num_cu * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(rsx_verify_kernel, dim3((uint32_t)blocks), dim3(256), 0, st,
                       static_cast<const uint8_t*>(d_data), (uint64_t)n, L->elem_bytes, L->key_offset, L->key_bytes,
                       L->key_kind, d_out);
    RSX_HIP(hipGetLastError());
    return RSX_OK;
} catch (...) {
    return RSX_ERR_HIP;
}

}  // extern "C"
