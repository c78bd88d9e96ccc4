// rsx_launch_impl.hpp -- definitions of the per-element-size launchers declared in
// rsx_internal.hpp.  Included only by rsx_es.hip, which instantiates them for ONE element size.
#pragma once
#include <cstdlib>
#include "rsx_internal.hpp"
#include "rsx_small_kernel.hpp"
#include "rsx_mid_kernels.hpp"

#ifndef RSX_HIST_BLOCKS_PER_CU
#define RSX_HIST_BLOCKS_PER_CU 8
#endif

namespace rsxh {

// ---- count phase of a first pass: J[r][v] for `digit` over the input regions ------------------
template <int ES, bool FLT>
int launch_hist_t(rsx_ctx* ctx, const void* src, const RegionGeom& g, const rsx_layout* L, uint32_t digit,
                  unsigned long long* J, unsigned long long* jclear, bool clear_status, hipStream_t st) {
    const uint64_t per_block = 512ull * 16;
    // the status words of the sweep that follows (its first half of the workspace) are zeroed by this kernel
    const uint64_t zero16_n = clear_status ? status_rows(g, ES) * RADIX * (status32(g) ? 4u : 8u) / 16u : 0u;
    uint64_t bpr = ((1ull << g.region_shift) + per_block - 1) / per_block;
    const uint64_t cap = ((uint64_t)ctx->num_cu * RSX_HIST_BLOCKS_PER_CU + g.num_regions - 1) / g.num_regions;
    if (bpr > cap) bpr = cap;
    if (bpr == 0) bpr = 1;
    LaunchTimer lt(ctx, RSX_PROF_HIST, st);
    hipLaunchKernelGGL((rsx_hist_kernel<ES, FLT>), dim3((uint32_t)(bpr * g.num_regions)), dim3(512), 0, st,
                       static_cast<const Elem<ES>*>(src), g, make_spec(L, digit), (uint32_t)bpr, J, jclear, status32(g) ? 1u : 0u,
                       static_cast<uint4*>(ctx->status), zero16_n, ctx->clean, ctx->gate, DigitSpec{}, nullptr, ctx->spec_dev);
    RSX_HIP(hipGetLastError());
    ctx->clean = CleanList{{nullptr, nullptr, nullptr}, {0, 0, 0}};  // done once per sort
    return RSX_OK;
}
template <int ES>
int launch_hist(rsx_ctx* ctx, const void* src, const RegionGeom& g, const rsx_layout* L, uint32_t digit,
                unsigned long long* J, unsigned long long* jclear, bool clear_status, hipStream_t st) {
    if (L->key_kind == RSX_KEY_FLOAT || (L->key_kind == RSX_KEY_SIGNED && (digit + 1 == L->key_bytes || ctx->spec_dev != nullptr)))
        return launch_hist_t<ES, true>(ctx, src, g, L, digit, J, jclear, clear_status, st);
    return launch_hist_t<ES, false>(ctx, src, g, L, digit, J, jclear, clear_status, st);
}

template <int ES>
int launch_hist2(rsx_ctx* ctx, const void* src, const RegionGeom& g, const rsx_layout* L, uint32_t digit,
                 unsigned long long* J, uint32_t digit2, unsigned long long* J2, unsigned long long* jclear, hipStream_t st) {
    const uint64_t per_block = 512ull * 16;
    const uint64_t zero16_n = status_rows(g, ES) * RADIX * (status32(g) ? 4u : 8u) / 16u;
    uint64_t bpr = ((1ull << g.region_shift) + per_block - 1) / per_block;
    const uint64_t cap = ((uint64_t)ctx->num_cu * RSX_HIST_BLOCKS_PER_CU + g.num_regions - 1) / g.num_regions;
    if (bpr > cap) bpr = cap;
    if (bpr == 0) bpr = 1;
    LaunchTimer lt(ctx, RSX_PROF_HIST, st);
    // one instantiation per key kind class: the general digit map is the identity for unsigned keys' specs
    if (L->key_kind != RSX_KEY_UNSIGNED)
        hipLaunchKernelGGL((rsx_hist_kernel<ES, true, true>), dim3((uint32_t)(bpr * g.num_regions)), dim3(512), 0, st,
                           static_cast<const Elem<ES>*>(src), g, make_spec(L, digit), (uint32_t)bpr, J, jclear, status32(g) ? 1u : 0u,
                           static_cast<uint4*>(ctx->status), zero16_n, ctx->clean, ctx->gate, make_spec(L, digit2), J2);
    else
        hipLaunchKernelGGL((rsx_hist_kernel<ES, false, true>), dim3((uint32_t)(bpr * g.num_regions)), dim3(512), 0, st,
                           static_cast<const Elem<ES>*>(src), g, make_spec(L, digit), (uint32_t)bpr, J, jclear, status32(g) ? 1u : 0u,
                           static_cast<uint4*>(ctx->status), zero16_n, ctx->clean, ctx->gate, make_spec(L, digit2), J2);
    RSX_HIP(hipGetLastError());
    ctx->clean = CleanList{{nullptr, nullptr, nullptr}, {0, 0, 0}};
    return RSX_OK;
}

// ---- scatter phase: one sweep pass -------------------------------------------------------------
// the per-dword masks of the signed/float key map (KeyXform in rsx_device.hpp)
inline KeyXform make_xform(const rsx_layout* L) {
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

template <int ES, typename S, int XF, bool NEXT, bool MID = false, bool STR = false>
int launch_sweep_t(rsx_ctx* ctx, const void* src, void* dst, const RegionGeom& g, const rsx_layout* L,
                   uint32_t digit, const unsigned long long* J, unsigned long long* jnext, unsigned long long* jzero,
                   hipStream_t st) {
    constexpr int KPT = kpt_for(ES);
    if (g.tile != (uint32_t)(wg_for(ES) * KPT)) return fail(ctx, RSX_ERR_INTERNAL, "launch_sweep: geometry of another tile size");
    constexpr int SWEEP_WG = wg_for(ES);
    constexpr int TILE = SWEEP_WG * KPT;
    const uint64_t rows = status_rows(g, ES);
    // Status words alternate between the two halves of the workspace.  The first half is zeroed by the
    // count kernel that precedes the first sweep; every pass zeroes, tile by tile, the half of the next.
    char* const half[2] = {static_cast<char*>(ctx->status), static_cast<char*>(ctx->status) + ctx->status_bytes};
    const uint32_t which = ctx->pass_index & 1u;
    (void)rows;
    SweepArgs a;
    a.status_clean = ctx->pass_last ? nullptr : half[which ^ 1u];
    a.src = src;
    a.dst = dst;
    a.g = g;
    a.J = J;
    a.status = half[which];
    a.tickets = ctx->tickets_override ? ctx->tickets_override : tickets_of(ctx, ctx->pass_index);
    a.prev_mode = ctx->pass_index ? tickets_of(ctx, ctx->pass_index - 1) + ROLL_SHARDS + 1 : nullptr;
    a.jnext = jnext;
    a.jzero = jzero;
    a.error = ctx->host_err_dev;
    a.spec = make_spec(L, digit);
    a.next = make_spec(L, NEXT ? digit + 1 : digit);
    a.spec.flip = a.next.flip = 0;  // the sweep sees mapped keys: plain digits
    a.xf = make_xform(L);
    a.tiles_per_region = (uint32_t)tiles_per_region(g, ES);
    a.opts = ((ctx->options & OPT_DYNAMIC_TILES) ? SWEEP_OPT_DYNAMIC : 0u) |
             ((ctx->options & OPT_NO_XCD_MAJOR) ? SWEEP_OPT_NO_XCD_MAJOR : 0u) |
             ((ctx->options & OPT_AGENT_STATUS) || !ctx->l2_local ? SWEEP_OPT_AGENT_STATUS : 0u) |
             ((ctx->options & OPT_RANK_CHECK) ? SWEEP_OPT_RANK_CHECK : 0u) |
             ((uint64_t)g.n * ES <= (2ull << 30) ? SWEEP_OPT_PREREAD : 0u);  // measured: a gain up to 2 GiB of data, a loss at 4 GiB
    a.dbg = ctx->dbg;
    a.mid_J = nullptr;
    a.mid_spec = a.spec;
    a.mid_cap = 0;
    a.gate = ctx->gate;
    a.spec_dev = ctx->spec_dev;
    a.mid_mode = ctx->pass_mid;
    a.mid_hint = ctx->host_err_dev + 8;  // second word group of the host-visible block
    if constexpr (MID) {
        a.mid_J = JT_of(ctx);
        a.mid_spec = make_spec(L, L->key_bytes - 1);
        a.mid_spec.flip = 0;
        a.mid_cap = bucket_cap(ES);
    }
    a.rank_atomic = (ctx->rank_atomic && !(ctx->options & OPT_BALLOT_RANKS)) ? 1u : 0u;
    a.hot_lanes = (ctx->options & OPT_ATOMIC_RANKS) ? 65u : ctx->hot_lanes;
    a.dbg_cnt = reinterpret_cast<unsigned long long*>(ctx->aux + OFF_DBG);
    const size_t lds = (size_t)TILE * ES + (SWEEP_WG / WAVE) * RADIX * ((RSX_WIDE_CNT && ES <= 4 && KPT >= 16 && SWEEP_WG <= 512) ? sizeof(uint32_t) : sizeof(uint16_t)) +
                       (NEXT ? (size_t)g.num_regions * RADIX * sizeof(uint32_t) : 0) + 128;
    auto kern = rsx_sweep_kernel<ES, KPT, SWEEP_WG, S, XF, NEXT, MID, STR>;
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
    // tiles of region r: every region ends with a tile of its own (partial unless the region length is a multiple
    // of the tile), so the total is NOT ceil(n / TILE)
    const uint32_t NR = g.num_regions;
    const uint64_t region_len = 1ull << g.region_shift;
    auto tiles_of = [&](uint32_t r) -> uint64_t {
        const uint64_t beg = (uint64_t)r << g.region_shift;
        const uint64_t len = g.n - beg < region_len ? g.n - beg : region_len;
        return (len + TILE - 1) / TILE;
    };
    uint64_t real_tiles = 0;
    for (uint32_t r = 0; r < NR; ++r) real_tiles += tiles_of(r);
    uint64_t grid = (uint64_t)ctx->num_cu * occ;
    if (grid > real_tiles) grid = real_tiles;
    if (grid < NR) grid = NR;
#ifdef RSX_TUNING
    if (const char* o = std::getenv("RSX_OCC")) {
        const long v = std::atol(o);
        if (v >= 1 && v <= 8) grid = (uint64_t)ctx->num_cu * (uint64_t)v;
    }
#endif
    if (grid > 0xFFFFu) grid = 0xFFFFu;  // wg_first entries are 16 bit
    {   // static mode: workgroups per region, proportional to the region's tile count, >= 1 each
        uint64_t cum = 0;
        for (uint32_t r = 0; r < NR; ++r) {
            a.wg_first[r] = (uint16_t)(cum * grid / real_tiles);
            cum += tiles_of(r);
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
        if (grid % 8 == 0 && !(ctx->options & OPT_NO_XCD_MAJOR))
            for (uint32_t r = 0; r < NR; ++r)
                if (a.wg_first[r] / (grid / 8) == (a.wg_first[r + 1] - 1u) / (grid / 8)) a.local_mask |= 1u << r;
    }
    if (ctx->options & OPT_VERBOSE) std::fprintf(stderr, "[rsx] sweep ES=%d NEXT=%d occ=%d grid=%llu lds=%zu tiles=%llu regions=%u\n", ES, (int)NEXT, occ, (unsigned long long)grid, lds, (unsigned long long)real_tiles, g.num_regions);
    LaunchTimer lt(ctx, RSX_PROF_SWEEP, st);
    hipLaunchKernelGGL(kern, dim3((uint32_t)grid), dim3(SWEEP_WG), lds, st, a);
    RSX_HIP(hipGetLastError());
    return RSX_OK;
}

template <int ES, typename S, int XF>
int launch_sweep_n(rsx_ctx* ctx, const void* src, void* dst, const RegionGeom& g, const rsx_layout* L,
                   uint32_t digit, const unsigned long long* J, unsigned long long* jnext, unsigned long long* jzero,
                   hipStream_t st) {
    if (ctx->spec_dev != nullptr) {  // the hybrid's two sweeps: window digits at any bit offset (the STR instantiation)
        if constexpr (ES >= 8 && (XF & 2) == 0) {
            if (jnext) return launch_sweep_t<ES, S, XF, true, false, true>(ctx, src, dst, g, L, digit, J, jnext, jzero, st);
            if constexpr (XF == 0) return launch_sweep_t<ES, S, 0, false, false, true>(ctx, src, dst, g, L, digit, J, nullptr, jzero, st);
        }
        return fail(ctx, RSX_ERR_INTERNAL, "launch_sweep: no kernel for this pass of the hybrid");
    }
    if constexpr ((XF & 2) == 0) {  // a pass that maps the keys back is a last pass: nothing to count for
        if constexpr (sizeof(S) == 4 && ES != 1) {  // the first sweep of a middle-size sort (regions of <= 2^30 elements by far)
            if (jnext && ctx->pass_mid != 0) return launch_sweep_t<ES, S, XF, true, true>(ctx, src, dst, g, L, digit, J, jnext, jzero, st);
        }
        if (jnext) return launch_sweep_t<ES, S, XF, true>(ctx, src, dst, g, L, digit, J, jnext, jzero, st);
    } else if (jnext) {
        return fail(ctx, RSX_ERR_ARG, "a last pass cannot count for a next one");
    }
    return launch_sweep_t<ES, S, XF, false>(ctx, src, dst, g, L, digit, J, nullptr, jzero, st);
}

template <int ES, typename S>
int launch_sweep_x(rsx_ctx* ctx, const void* src, void* dst, const RegionGeom& g, const rsx_layout* L,
                   uint32_t digit, const unsigned long long* J, unsigned long long* jnext, unsigned long long* jzero,
                   int xf, hipStream_t st) {
    switch (L->key_kind == RSX_KEY_UNSIGNED ? 0 : xf) {
        case 1: return launch_sweep_n<ES, S, 1>(ctx, src, dst, g, L, digit, J, jnext, jzero, st);
        case 2: return launch_sweep_n<ES, S, 2>(ctx, src, dst, g, L, digit, J, jnext, jzero, st);
        case 3: return launch_sweep_n<ES, S, 3>(ctx, src, dst, g, L, digit, J, jnext, jzero, st);
        default: return launch_sweep_n<ES, S, 0>(ctx, src, dst, g, L, digit, J, jnext, jzero, st);
    }
}

template <int ES>
int launch_sweep(rsx_ctx* ctx, const void* src, void* dst, const RegionGeom& g, const rsx_layout* L, uint32_t digit,
                 const unsigned long long* J, unsigned long long* jnext, unsigned long long* jzero, int xf,
                 hipStream_t st) {
    if (status32(g)) return launch_sweep_x<ES, uint32_t>(ctx, src, dst, g, L, digit, J, jnext, jzero, xf, st);
    return launch_sweep_x<ES, uint64_t>(ctx, src, dst, g, L, digit, J, jnext, jzero, xf, st);
}

// ---- arrays of at most one tile: all passes in one launch of one workgroup ----------------------
template <int ES>
int launch_small_sort(rsx_ctx* ctx, void* data, size_t n, const rsx_layout* L, hipStream_t st) {
    constexpr int KPT = kpt_for(ES);
    if (n == 0 || n > (size_t)512 * KPT || L->key_bytes > 16) return fail(ctx, RSX_ERR_INTERNAL, "launch_small_sort: size out of range");
    SmallArgs a;
    std::memset(&a, 0, sizeof a);
    a.src = data;
    a.data = data;
    a.n = (uint32_t)n;
    a.passes = L->key_bytes;
    a.rank_atomic = (ctx->rank_atomic && !(ctx->options & OPT_BALLOT_RANKS)) ? 1u : 0u;
    a.map_load = a.map_store = L->key_kind == RSX_KEY_UNSIGNED ? 0u : 1u;
    for (uint32_t d = 0; d < L->key_bytes; ++d) {
        a.spec[d] = make_spec(L, d);
        a.spec[d].flip = 0;  // the kernel sees mapped keys: plain digits
    }
    a.xf = make_xform(L);
    const size_t lds = (size_t)512 * KPT * ES + 8 * RADIX * sizeof(uint32_t) + 64;
    auto kern = rsx_small_sort_kernel<ES, KPT>;
    LaunchTimer lt(ctx, RSX_PROF_OTHER, st);
    hipLaunchKernelGGL(kern, dim3(1), dim3(512), lds, st, a);
    RSX_HIP(hipGetLastError());
    return RSX_OK;
}

// ---- middle-size sorts: split by the top digit (count -> scan -> scatter, one launch each) --------
inline uint32_t* mid_totals_of(rsx_ctx* ctx) { return reinterpret_cast<uint32_t*>(ctx->aux + OFF_BASE); }
template <int ES>
int launch_mid_split(rsx_ctx* ctx, const void* src, void* dst, size_t n, const rsx_layout* L, hipStream_t st) {
    if constexpr (ES == 1) {
        return fail(ctx, RSX_ERR_INTERNAL, "launch_mid_split: one-byte elements");
    } else {
        constexpr int KPT = mid_kpt_for(ES);
        constexpr uint32_t TILE = 512u * KPT;
        MidArgs a;
        std::memset(&a, 0, sizeof a);
        a.src = src;
        a.dst = dst;
        a.n = (uint32_t)n;
        a.tiles = (uint32_t)((n + TILE - 1) / TILE);
        if ((size_t)a.tiles * RADIX * sizeof(uint32_t) > ctx->status_bytes) return fail(ctx, RSX_ERR_INTERNAL, "launch_mid_split: workspace");
        a.C = static_cast<uint32_t*>(ctx->status);
        a.X = reinterpret_cast<uint32_t*>(static_cast<char*>(ctx->status) + ctx->status_bytes);
        a.T = mid_totals_of(ctx);
        a.spec = make_spec(L, L->key_bytes - 1);  // count: the raw key's top byte through the digit map
        a.xf = make_xform(L);
        a.map_keys = L->key_kind == RSX_KEY_UNSIGNED ? 0u : 1u;
        a.rank_atomic = (ctx->rank_atomic && !(ctx->options & OPT_BALLOT_RANKS)) ? 1u : 0u;
        {
            LaunchTimer lt(ctx, RSX_PROF_HIST, st);
            if (L->key_kind != RSX_KEY_UNSIGNED) hipLaunchKernelGGL((rsx_tilecount_kernel<ES, KPT, true>), dim3(a.tiles), dim3(512), 0, st, a);
            else hipLaunchKernelGGL((rsx_tilecount_kernel<ES, KPT, false>), dim3(a.tiles), dim3(512), 0, st, a);
            RSX_HIP(hipGetLastError());
        }
        {
            LaunchTimer lt(ctx, RSX_PROF_SCAN, st);
            hipLaunchKernelGGL(rsx_tilescan_kernel<ES>, dim3(RADIX), dim3(256), 0, st, a);
            RSX_HIP(hipGetLastError());
        }
        a.spec.flip = 0;  // scatter: the keys are mapped on load, plain digits from there on
        const size_t lds = (size_t)TILE * ES + 8 * RADIX * sizeof(uint32_t) + 64 + 2 * RADIX * sizeof(uint32_t);
        LaunchTimer lt(ctx, RSX_PROF_OTHER, st);
        hipLaunchKernelGGL((rsx_tilescatter_kernel<ES, KPT>), dim3(a.tiles), dim3(512), lds, st, a);
        RSX_HIP(hipGetLastError());
        return RSX_OK;
    }
}

// ---- ... then the 256 buckets, each sorted by one workgroup --------------------------------------
// the bucket kernels start their LDS passes at the digit their bucket's size allows (RSX_OPT_BUCKET_SKIP, first_digit_for)
// and compare neighbours on the key bytes from there up
inline void set_skip_mask(rsx_ctx* ctx, SmallArgs& a, const rsx_layout* L) {
    a.no_skip = ctx->bucket_no_skip;
    a.key_offset = L->key_offset;  // (the kernel builds its compare masks from these)
    a.key_bytes = L->key_bytes;
}

template <int ES>
int launch_bucket_sort(rsx_ctx* ctx, const void* src, void* dst, const RegionGeom& g, const rsx_layout* L, hipStream_t st) {
    if constexpr (ES == 1) {
        return fail(ctx, RSX_ERR_INTERNAL, "launch_bucket_sort: one-byte elements");
    } else {
        constexpr int KPT = bucket_kpt_for(ES);
        if (L->key_bytes < 2 || L->key_bytes > 16) return fail(ctx, RSX_ERR_INTERNAL, "launch_bucket_sort: key width out of range");
        SmallArgs a;
        std::memset(&a, 0, sizeof a);
        a.src = src;
        a.data = dst;
        a.passes = L->key_bytes - 1;
        a.rank_atomic = (ctx->rank_atomic && !(ctx->options & OPT_BALLOT_RANKS)) ? 1u : 0u;
        a.map_load = 0;  // the first sweep mapped the keys
        a.map_store = L->key_kind == RSX_KEY_UNSIGNED ? 0u : 1u;
        for (uint32_t d = 0; d + 1 < L->key_bytes; ++d) {
            a.spec[d] = make_spec(L, d);
            a.spec[d].flip = 0;
        }
        a.xf = make_xform(L);
        (void)g;
        a.top_tot = mid_totals_of(ctx);
        a.cap = bucket_cap(ES);
        a.hint = ctx->host_err_dev + 8;
        set_skip_mask(ctx, a, L);
        LaunchTimer lt(ctx, RSX_PROF_OTHER, st);
        if (ctx->bucket_small) {  // small buckets, all known to fit: 256 threads each
            const size_t lds = (size_t)cape<ES, KPT, 256>() * ES + 4 * RADIX * bucket_cnt_bytes(ES) + 64 + 3 * RADIX * sizeof(uint32_t);
            hipLaunchKernelGGL((rsx_bucket_sort_kernel<ES, KPT, 256>), dim3(RADIX), dim3(256), lds, st, a);
        } else {
            const size_t lds = (size_t)cape<ES, KPT, 1024>() * ES + 16 * RADIX * bucket_cnt_bytes(ES) + 64 + 3 * RADIX * sizeof(uint32_t);
            auto kern = rsx_bucket_sort_kernel<ES, KPT, 1024>;
            ensure_lds(ctx, reinterpret_cast<const void*>(kern), lds);
            hipLaunchKernelGGL(kern, dim3(RADIX), dim3(1024), lds, st, a);
        }
        RSX_HIP(hipGetLastError());
        return RSX_OK;
    }
}

// ---- wide keys, large arrays: count of the top 16 bits; the buckets sorted in LDS -------------------
// the hybrid's plan: which 16 bits of the mapped key the array is partitioned by (rsx_wideplan_kernel)
template <int ES>
int launch_wideplan(rsx_ctx* ctx, const void* src, size_t n, const rsx_layout* L, WidePlan* plan, hipStream_t st) {
    if constexpr (ES < 4) {
        return fail(ctx, RSX_ERR_INTERNAL, "launch_wideplan: narrow elements");
    } else {
        LaunchTimer lt(ctx, RSX_PROF_SCAN, st);
        if (L->key_kind != RSX_KEY_UNSIGNED)
            hipLaunchKernelGGL((rsx_wideplan_kernel<ES, true>), dim3(WIDEPLAN_BLOCKS), dim3(1024), 0, st, static_cast<const Elem<ES>*>(src), (uint64_t)n, L->key_offset,
                               L->key_bytes, L->key_kind, make_xform(L), plan);
        else
            hipLaunchKernelGGL((rsx_wideplan_kernel<ES, false>), dim3(WIDEPLAN_BLOCKS), dim3(1024), 0, st, static_cast<const Elem<ES>*>(src), (uint64_t)n, L->key_offset,
                               L->key_bytes, L->key_kind, make_xform(L), plan);
        RSX_HIP(hipGetLastError());
        return RSX_OK;
    }
}

template <int ES>
int launch_count16top(rsx_ctx* ctx, const void* src, size_t n, const rsx_layout* L, WidePlan* plan, uint32_t* P, uint32_t parts,
                      uint32_t region_shift, uint32_t k, hipStream_t st) {
    if constexpr (ES < 4) {
        return fail(ctx, RSX_ERR_INTERNAL, "launch_count16top: narrow elements");
    } else {
        LaunchTimer lt(ctx, RSX_PROF_HIST, st);
        if (L->key_kind != RSX_KEY_UNSIGNED) {
            auto kern = rsx_count16top_kernel<ES, true>;
            ensure_lds(ctx, reinterpret_cast<const void*>(kern), 131072);
            hipLaunchKernelGGL(kern, dim3(parts), dim3(1024), 131072, st, static_cast<const Elem<ES>*>(src), (uint64_t)n, plan, make_xform(L), P,
                               ctx->ovf16, region_shift, k);
        } else {
            auto kern = rsx_count16top_kernel<ES, false>;
            ensure_lds(ctx, reinterpret_cast<const void*>(kern), 131072);
            hipLaunchKernelGGL(kern, dim3(parts), dim3(1024), 131072, st, static_cast<const Elem<ES>*>(src), (uint64_t)n, plan, make_xform(L), P,
                               ctx->ovf16, region_shift, k);
        }
        RSX_HIP(hipGetLastError());
        return RSX_OK;
    }
}

// the first sweep's count matrix from the counters of launch_count16top (k chunks per region), + the side jobs of launch_hist
template <int ES>
int launch_marginal16(rsx_ctx* ctx, const uint32_t* P, uint32_t parts, uint32_t k, const RegionGeom& g, unsigned long long* J,
                      unsigned long long* jclear, hipStream_t st) {
    const uint64_t zero16_n = status_rows(g, ES) * RADIX * (status32(g) ? 4u : 8u) / 16u;
    uint32_t grid = (uint32_t)ctx->num_cu * 8u;
    if (grid < parts) grid = parts;
    LaunchTimer lt(ctx, RSX_PROF_HIST, st);
    hipLaunchKernelGGL(rsx_marginal16_kernel<ES>, dim3(grid), dim3(256), 0, st, P, parts, k, g, J, jclear, status32(g) ? 1u : 0u,
                       static_cast<uint4*>(ctx->status), zero16_n, ctx->clean, ctx->gate);
    RSX_HIP(hipGetLastError());
    ctx->clean = CleanList{{nullptr, nullptr, nullptr}, {0, 0, 0}};
    return RSX_OK;
}

// groups of 2^gs small buckets (0: none): about 3/4 of what a 512-thread workgroup of the bucket kernel holds
template <int ES>
uint32_t bucket16_group_shift(rsx_ctx* ctx, size_t n, const rsx_layout* L) {
    constexpr int KPT = bucket_kpt_for(ES);
    const uint64_t avg = (uint64_t)n / 65536u;
    uint32_t gs = 0;
    if (L->key_bytes >= 8 && ctx->bucket_group)
        while (gs < 6 && (avg << (gs + 1)) <= (uint64_t)512 * KPT * 3 / 4) ++gs;
    return gs >= 2 ? gs : 0;
}

template <int ES>
int launch_bucket16(rsx_ctx* ctx, void* data, void* scratch, size_t n, const rsx_layout* L, const uint64_t* starts, const WidePlan* plan,
                    hipStream_t st) {
    if constexpr (ES < 4) {
        return fail(ctx, RSX_ERR_INTERNAL, "launch_bucket16: narrow elements");
    } else {
        constexpr int KPT = bucket_kpt_for(ES);
        if (L->key_bytes < 4) return fail(ctx, RSX_ERR_INTERNAL, "launch_bucket16: key width out of range");
        SmallArgs a;
        std::memset(&a, 0, sizeof a);
        a.src = data;
        a.data = data;
        a.passes = L->key_bytes - 2;  // (what the sort THROUGH MEMORY of an oversized bucket runs: an even number, every digit the
                                      // LDS passes could need; the LDS passes themselves follow the device's plan)
        a.rank_atomic = (ctx->rank_atomic && !(ctx->options & OPT_BALLOT_RANKS)) ? 1u : 0u;
        a.map_load = 0;  // the first sweep mapped the keys
        a.map_store = L->key_kind == RSX_KEY_UNSIGNED ? 0u : 1u;
        for (uint32_t d = 0; d < L->key_bytes; ++d) {
            a.spec[d] = make_spec(L, d);
            a.spec[d].flip = 0;
        }
        a.xf = make_xform(L);
        a.cap = bucket_cap(ES);
        a.no_skip = ctx->bucket_no_skip;
        a.key_offset = L->key_offset;
        a.key_bytes = L->key_bytes;
        LaunchTimer lt(ctx, RSX_PROF_OTHER, st);
        // Every form the device's verdict can name is enqueued behind its own gate (rsx_scan16_kernel picks the smallest
        // workgroup that holds all but a handful of THIS input's buckets; a form whose workgroup cannot even hold the
        // average bucket is not enqueued): 256, 512 or 1024 threads x KPT elements, as many workgroups per CU as LDS and
        // registers allow (3, 2, 1); small buckets in groups of 2^gs, about 3/4 of what a 512-thread workgroup holds,
        // sorted by all digits up to the window's top (keys of at least 8 bytes).
        const uint64_t avg = (uint64_t)n / 65536u;
        const uint32_t gs = bucket16_group_shift<ES>(ctx, n, L);
        const Gate base = ctx->gate;  // (null word: no gates -- never the case for this kernel)
        auto go = [&](auto wgc, auto kc, uint32_t form, uint32_t group_shift) {
            constexpr int WGS = decltype(wgc)::value;
            constexpr int K = decltype(kc)::value;
            SmallArgs b = a;
            b.group_shift = group_shift;
            if (group_shift) b.passes = L->key_bytes;
            static_assert(cape<ES, K, WGS>() == bucket_cape(ES, K, WGS), "host and device agree on what a workgroup holds");
            const size_t lds = (size_t)cape<ES, K, WGS>() * ES + (WGS / 64) * RADIX * bucket_cnt_bytes(ES) + 64 + 3 * RADIX * sizeof(uint32_t);
            auto kern = rsx_bucket16_kernel<ES, K, WGS>;
            ensure_lds(ctx, reinterpret_cast<const void*>(kern), lds);
            int per_cu = (int)((size_t)163840 / lds);
            if (per_cu < 1) per_cu = 1;
            if (per_cu > 4 * RSX_B16_WAVES(WGS) * 64 / WGS) per_cu = 4 * RSX_B16_WAVES(WGS) * 64 / WGS;
            const Gate g{base.word, VERDICT_PATH_MASK | VERDICT_FORM_MASK, VERDICT_HYBRID | form};
            hipLaunchKernelGGL(kern, dim3((uint32_t)(ctx->num_cu * per_cu)), dim3(WGS), lds, st, b, starts, scratch, plan, g);
        };
        using std::integral_constant;
        constexpr int KBIG = wide_kpt_for(ES);  // (the longer form only where the average bucket needs it: wide_big_form)
        if (gs >= 2) go(integral_constant<int, 512>{}, integral_constant<int, KPT>{}, VERDICT_GROUPS, gs);
        if (avg <= (uint64_t)256 * KPT) go(integral_constant<int, 256>{}, integral_constant<int, KPT>{}, VERDICT_WG256, 0);
        if (avg <= (uint64_t)512 * KPT) go(integral_constant<int, 512>{}, integral_constant<int, KPT>{}, VERDICT_WG512, 0);
        const bool big_form = KBIG != KPT && wide_big_form(ES, n);
        if (big_form) go(integral_constant<int, 1024>{}, integral_constant<int, KBIG>{}, VERDICT_WG1024, 0);
        else go(integral_constant<int, 1024>{}, integral_constant<int, KPT>{}, VERDICT_WG1024, 0);
        {   // the buckets above the chosen form's workgroup (VERDICT_MEDIUM): one workgroup of the largest kind each
            constexpr int KMED = medium_kpt_for(ES);
            const size_t lds = (size_t)cape<ES, KMED, 1024>() * ES + 16 * RADIX * bucket_cnt_bytes(ES) + 64 + 3 * RADIX * sizeof(uint32_t);
            auto kern = rsx_bucket16_medium_kernel<ES, KMED, 1024>;
            ensure_lds(ctx, reinterpret_cast<const void*>(kern), lds);
            const Gate g{base.word, VERDICT_PATH_MASK | VERDICT_MEDIUM, VERDICT_HYBRID | VERDICT_MEDIUM};
            hipLaunchKernelGGL(kern, dim3((uint32_t)ctx->num_cu), dim3(1024), lds, st, a, starts, scratch, plan, (uint32_t)(256 * KPT), (uint32_t)(512 * KPT),
                               (uint32_t)(big_form ? cape<ES, KBIG, 1024>() : cape<ES, KPT, 1024>()), g);
        }
        RSX_HIP(hipGetLastError());
        return RSX_OK;
    }
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

}  // namespace rsxh
