/*
 * rsx_oracle.c -- CPU restatement of jgrodzki/radix_sort's LSD radix sort.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under radix_sort_amd/ (the product) may
 * include, link, import or execute this file; only tests/, bench.py's
 * `cpu_baseline` leg and __graft_entry__.smoke() use it, and only as the
 * checker / the timed CPU baseline, never as the thing shipped.
 *
 * What it restates (file:line relative to the reference checkout):
 *   - key -> digit maps ............ src/radix_sort/radix_digits.rs:7-136
 *   - single-thread LSD ............ src/radix_sort/mod.rs:183-212 (radix_sort0)
 *   - thread-parallel LSD .......... src/radix_sort/mod.rs:61-176  (radix_sort)
 *       chunking                     mod.rs:66-70
 *       temp alloc + page touch      mod.rs:71-82
 *       ping-pong                    mod.rs:84-89
 *       count                        mod.rs:90-109
 *       digit-major/chunk-minor scan mod.rs:110-120
 *       96-element buffered scatter  mod.rs:121-168
 *       odd-D copy-back              mod.rs:170-174
 *
 * Pinning: the reference ships NO golden vectors (tests.rs draws from an
 * unseeded thread_rng); what its tests pin is a property with a unique answer
 * -- output == stable sort by mapped key (tests.rs:7-23,133-187).  This oracle
 * is checked against that property through an independent implementation
 * (numpy stable argsort on the mapped key) in tests/test_oracle.py, and the
 * fixtures in tests/golden/ are produced by both and required to agree.
 *
 * Elements are opaque byte strings of `elem_bytes`; the key is the
 * little-endian integer of `key_bytes` at `key_offset` (Rust primitives on the
 * x86-64/little-endian targets the reference runs on; tuples `(T,U)` have the
 * key `.0` at a run-time offset -- radix_digits.rs:126-136).
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { ORC_UNSIGNED = 0, ORC_SIGNED = 1, ORC_FLOAT = 2 };

typedef struct {
    uint32_t elem_bytes;
    uint32_t key_offset;
    uint32_t key_bytes; /* 1,2,4,8,16 == NUMBER_OF_DIGITS */
    uint32_t key_kind;
} orc_layout;

/* radix_digits.rs: get_digit(index).
 *   unsigned (7-53):  (x >> 8*index) as u8                -> byte `index`
 *   signed  (55-101): ((x ^ MIN) >> 8*index) as u8        -> top byte ^ 0x80
 *   float  (103-124): b ^= (b >> (bits-1)) | MIN; byte    -> negative: every
 *                     byte ^ 0xFF, non-negative: top byte ^ 0x80
 */
static inline uint32_t orc_digit(const uint8_t *e, const orc_layout *L, uint32_t index) {
    const uint8_t *k = e + L->key_offset;
    uint32_t top = L->key_bytes - 1;
    uint8_t b = k[index];
    switch (L->key_kind) {
    case ORC_SIGNED:
        if (index == top) b ^= 0x80;
        break;
    case ORC_FLOAT:
        if (k[top] & 0x80) b ^= 0xFF;
        else if (index == top) b ^= 0x80;
        break;
    default:
        break;
    }
    return b;
}

uint32_t orc_get_digit(const void *elem, const orc_layout *L, uint32_t index) {
    return orc_digit((const uint8_t *)elem, L, index);
}

/* mapped key of one element as `key_bytes` little-endian bytes (order-preserving
 * as an unsigned integer) -- used by the numpy cross-check. */
void orc_map_keys(const void *data, size_t n, const orc_layout *L, void *out) {
    const uint8_t *p = (const uint8_t *)data;
    uint8_t *o = (uint8_t *)out;
    for (size_t i = 0; i < n; ++i)
        for (uint32_t d = 0; d < L->key_bytes; ++d)
            o[i * L->key_bytes + d] = (uint8_t)orc_digit(p + i * L->elem_bytes, L, d);
}

/* ---- mod.rs:183-212  radix_sort0: single thread ------------------------- */
int orc_radix_sort0(void *data, size_t n, const orc_layout *L) {
    const size_t s = L->elem_bytes;
    if (n == 0) return 0;
    uint8_t *self = (uint8_t *)data;
    uint8_t *temp = (uint8_t *)malloc(n * s); /* vec![T::default(); len] */
    if (!temp) return -1;
    for (uint32_t d = 0; d < L->key_bytes; ++d) {
        const uint8_t *src = (d % 2 == 0) ? self : temp; /* mod.rs:186-190 */
        uint8_t *dst = (d % 2 == 0) ? temp : self;
        size_t hist[256] = {0};
        for (size_t i = 0; i < n; ++i) hist[orc_digit(src + i * s, L, d)]++; /* :192-194 */
        size_t start = 0; /* fold exclusive scan :195-202 */
        for (int v = 0; v < 256; ++v) {
            size_t c = hist[v];
            hist[v] = start;
            start += c;
        }
        for (size_t i = 0; i < n; ++i) { /* :203-207 */
            uint32_t v = orc_digit(src + i * s, L, d);
            memcpy(dst + hist[v] * s, src + i * s, s);
            hist[v]++;
        }
    }
    if (L->key_bytes % 2 == 1) memcpy(self, temp, n * s); /* :209-211 */
    free(temp);
    return 0;
}

/* ---- mod.rs:61-176  radix_sort: thread-parallel -------------------------- */
#define ORC_BUFFER_SIZE 96 /* mod.rs:64 */
#define ORC_PAGE_SIZE 4096 /* mod.rs:65 */

typedef struct {
    const uint8_t *src;
    uint8_t *dst;
    size_t begin, end; /* element range of this chunk */
    const orc_layout *L;
    uint32_t digit;
    size_t hist[256]; /* count out / bin_starts in */
    uint8_t *derand;  /* 256 * 96 * s bytes, thread-private (mod.rs:126-129) */
} orc_job;

/* Bodies are always-inlined with a constant element size for the common sizes so the
 * per-element memcpy becomes a register move (the Rust original is monomorphised per T). */
static inline __attribute__((always_inline)) void orc_count_body(orc_job *j, const size_t s) { /* mod.rs:94-100 */
    memset(j->hist, 0, sizeof j->hist);
    for (size_t i = j->begin; i < j->end; ++i) j->hist[orc_digit(j->src + i * s, j->L, j->digit)]++;
}

static inline __attribute__((always_inline)) void orc_scatter_body(orc_job *j, const size_t s) { /* mod.rs:125-166 */
    size_t sizes[256] = {0};
    for (size_t i = j->begin; i < j->end; ++i) {
        const uint8_t *e = j->src + i * s;
        uint32_t v = orc_digit(e, j->L, j->digit);
        memcpy(j->derand + ((size_t)v * ORC_BUFFER_SIZE + sizes[v]) * s, e, s); /* :133-141 */
        if (++sizes[v] == ORC_BUFFER_SIZE) {                                     /* :142-153 */
            memcpy(j->dst + j->hist[v] * s, j->derand + (size_t)v * ORC_BUFFER_SIZE * s,
                   ORC_BUFFER_SIZE * s);
            j->hist[v] += ORC_BUFFER_SIZE;
            sizes[v] = 0;
        }
    }
    for (int v = 0; v < 256; ++v) /* :155-165 */
        if (sizes[v] > 0)
            memcpy(j->dst + j->hist[v] * s, j->derand + (size_t)v * ORC_BUFFER_SIZE * s, sizes[v] * s);
}

#define ORC_BY_SIZE(BODY, j)                \
    switch ((j)->L->elem_bytes) {           \
    case 1: BODY(j, 1); break;              \
    case 2: BODY(j, 2); break;              \
    case 4: BODY(j, 4); break;              \
    case 8: BODY(j, 8); break;              \
    case 16: BODY(j, 16); break;            \
    default: BODY(j, (j)->L->elem_bytes);   \
    }

static void *orc_count_worker(void *arg) {
    orc_job *j = (orc_job *)arg;
    ORC_BY_SIZE(orc_count_body, j)
    return NULL;
}

static void *orc_scatter_worker(void *arg) {
    orc_job *j = (orc_job *)arg;
    ORC_BY_SIZE(orc_scatter_body, j)
    return NULL;
}

/* `threads` plays available_parallelism() (mod.rs:66-70).  The reference panics
 * on an empty slice (chunks(0)); there is no output to compare, so n == 0 is a
 * no-op here.  Returns 0, or -1 on allocation/thread failure. */
int orc_radix_sort(void *data, size_t n, const orc_layout *L, int threads) {
    const size_t s = L->elem_bytes;
    if (n == 0) return 0;
    if (threads < 1) threads = 1;
    const size_t per = (n + (size_t)threads - 1) / (size_t)threads; /* div_ceil :66-70 */
    const size_t chunks = (n + per - 1) / per;                      /* src.chunks(per) */
    uint8_t *self = (uint8_t *)data;
    uint8_t *temp = (uint8_t *)malloc(n * s); /* uninitialised, :71-73 */
    orc_job *jobs = (orc_job *)calloc(chunks, sizeof(orc_job));
    pthread_t *tids = (pthread_t *)calloc(chunks, sizeof(pthread_t));
    uint8_t *derand = (uint8_t *)malloc(chunks * 256 * ORC_BUFFER_SIZE * s);
    int rc = 0;
    if (!temp || !jobs || !tids || !derand) {
        rc = -1;
        goto out;
    }
    for (size_t b = 0; b < n * s; b += ORC_PAGE_SIZE) temp[b] = 0; /* page touch :74-82 */
    for (size_t c = 0; c < chunks; ++c) {
        jobs[c].begin = c * per;
        jobs[c].end = (c + 1) * per < n ? (c + 1) * per : n;
        jobs[c].L = L;
        jobs[c].derand = derand + c * 256 * ORC_BUFFER_SIZE * s;
    }
    for (uint32_t d = 0; d < L->key_bytes; ++d) { /* :84 */
        const uint8_t *src = (d % 2 == 0) ? self : temp; /* :85-89 */
        uint8_t *dst = (d % 2 == 0) ? temp : self;
        for (size_t c = 0; c < chunks; ++c) {
            jobs[c].src = src;
            jobs[c].dst = dst;
            jobs[c].digit = d;
        }
        /* count: one OS thread per chunk, spawned per pass (:90-109) */
        if (chunks == 1) orc_count_worker(&jobs[0]);
        else {
            for (size_t c = 0; c < chunks; ++c)
                if (pthread_create(&tids[c], NULL, orc_count_worker, &jobs[c])) {
                    for (size_t k = 0; k < c; ++k) pthread_join(tids[k], NULL);
                    rc = -1;
                    goto out;
                }
            for (size_t c = 0; c < chunks; ++c) pthread_join(tids[c], NULL);
        }
        /* prefix: digit-major, chunk-minor exclusive running sum (:110-120) */
        size_t prefix = 0;
        for (int v = 0; v < 256; ++v)
            for (size_t c = 0; c < chunks; ++c) {
                size_t cnt = jobs[c].hist[v];
                jobs[c].hist[v] = prefix;
                prefix += cnt;
            }
        /* scatter through 96-element write-combining buffers (:121-168) */
        if (chunks == 1) orc_scatter_worker(&jobs[0]);
        else {
            for (size_t c = 0; c < chunks; ++c)
                if (pthread_create(&tids[c], NULL, orc_scatter_worker, &jobs[c])) {
                    for (size_t k = 0; k < c; ++k) pthread_join(tids[k], NULL);
                    rc = -1;
                    goto out;
                }
            for (size_t c = 0; c < chunks; ++c) pthread_join(tids[c], NULL);
        }
    }
    if (L->key_bytes % 2 == 1) memcpy(self, temp, n * s); /* :170-174 */
out:
    free(derand);
    free(tids);
    free(jobs);
    free(temp);
    return rc;
}

/* One LSD pass (count -> scan -> scatter of digit `d`) src -> dst, single
 * thread: the unit the multi-GPU driver's per-pass bucket exchange is checked
 * against (mod.rs:191-207 for one value of current_digit_index). */
int orc_partition_pass(const void *src_, void *dst_, size_t n, const orc_layout *L, uint32_t d,
                       uint64_t *hist_out /* [256] counts, may be NULL */) {
    const size_t s = L->elem_bytes;
    const uint8_t *src = (const uint8_t *)src_;
    uint8_t *dst = (uint8_t *)dst_;
    size_t hist[256] = {0};
    for (size_t i = 0; i < n; ++i) hist[orc_digit(src + i * s, L, d)]++;
    size_t start = 0;
    for (int v = 0; v < 256; ++v) {
        size_t c = hist[v];
        if (hist_out) hist_out[v] = c;
        hist[v] = start;
        start += c;
    }
    for (size_t i = 0; i < n; ++i) {
        uint32_t v = orc_digit(src + i * s, L, d);
        memcpy(dst + hist[v] * s, src + i * s, s);
        hist[v]++;
    }
    return 0;
}

/* Counter-based input generator shared with the GPU side (SplitMix64 of
 * seed + index): lets CPU and GPU build identical arrays without PCIe.
 * Harness-side only (replaces thread_rng of main.rs:27-30 / tests.rs). */
static inline uint64_t orc_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
uint64_t orc_rand64(uint64_t seed, uint64_t index) { return orc_splitmix64(seed + index * 0x9E3779B97F4A7C15ull); }

/* ---- input generators ---------------------------------------------------------
 * CPU restatement of the library's on-device generators (rsx_generate_device), which in turn
 * give the SHAPES of the reference's distributions (src/distr.rs: KeyUniform :40-52, Zipf
 * :54-76,108-130, step-uniform :78-106,132-160, geometric MyExp :3-38, `(key, 0)` pairs
 * :22-26) -- the reference draws from rand_distr with an unseeded thread_rng, so there is no
 * stream of its own to reproduce.  Integer arithmetic only: the same (seed, index) must give the
 * same bytes here and on the device (tests/test_generators.py).
 *   2^(2^-i) as 1.63 fixed point, i = 1..32 */
static const uint64_t ORC_EXP2_TAB[32] = {
    0xB504F333F9DE6484ull, 0x9837F0518DB8A96Full, 0x8B95C1E3EA8BD6E7ull, 0x85AAC367CC487B15ull,
    0x82CD8698AC2BA1D7ull, 0x8164D1F3BC030773ull, 0x80B1ED4FD999AB6Cull, 0x8058D7D2D5E5F6B1ull,
    0x802C6436D0E04F51ull, 0x8016302F17467628ull, 0x800B179C82028FD1ull, 0x80058BAF7FEE3B5Dull,
    0x8002C5D00FDCFCB7ull, 0x800162E61BED4A49ull, 0x8000B17292F702A4ull, 0x800058B92ABBAE02ull,
    0x80002C5C8DADE4D7ull, 0x8000162E44EAF636ull, 0x80000B1721FA7C19ull, 0x8000058B90DE7E4Dull,
    0x800002C5C8678F37ull, 0x80000162E431DBA0ull, 0x800000B1721872D1ull, 0x80000058B90C1AA9ull,
    0x8000002C5C8605A4ull, 0x800000162E4300E6ull, 0x8000000B17217FF8ull, 0x800000058B90BFDDull,
    0x80000002C5C85FE7ull, 0x8000000162E42FF2ull, 0x80000000B17217F8ull, 0x8000000058B90BFCull};

static inline uint64_t orc_mulhi64(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }

/* floor(2^(e + f/2^32)), 0 <= e <= 63 */
static uint64_t orc_exp2_floor(uint32_t e, uint32_t f) {
    uint64_t m = 1ull << 63;
    for (int i = 0; i < 32; ++i)
        if (f & (0x80000000u >> i)) m = orc_mulhi64(m, ORC_EXP2_TAB[i]) << 1;
    return m >> (63u - e);
}
/* -log2(w / 2^64), w >= 1, as 32.32 fixed point */
static uint64_t orc_neg_log2(uint64_t w) {
    const uint32_t lz = (uint32_t)__builtin_clzll(w);
    uint64_t m = w << lz;
    uint32_t frac = 0;
    for (int i = 0; i < 32; ++i) {
        m = orc_mulhi64(m, m);
        if (m >> 63) frac |= 0x80000000u >> i;
        else m <<= 1;
    }
    return (64ull << 32) - ((((uint64_t)(63u - lz)) << 32) | frac);
}

enum { ORC_GEN_UNIFORM = 0, ORC_GEN_ZIPF = 1, ORC_GEN_STEP = 2, ORC_GEN_SORTED = 3, ORC_GEN_REVERSED = 4,
       ORC_GEN_CONSTANT = 5, ORC_GEN_GEOMETRIC = 6 };

/* iparam: STEP: number of values; CONSTANT: the value; GEOMETRIC: -log2(1 - p) as 32.32 fixed point
 * (the caller makes it from p with the same expression the library uses).  ZIPF: exponent 1 only. */
int orc_generate(void *data, size_t n, const orc_layout *L, int gen, uint64_t seed, uint64_t iparam,
                 uint64_t index_base, int payload_zero) {
    uint8_t *p = (uint8_t *)data;
    const uint32_t bits = L->key_bytes * 8;
    if (gen < ORC_GEN_UNIFORM || gen > ORC_GEN_GEOMETRIC) return -1;
    if ((gen == ORC_GEN_STEP || gen == ORC_GEN_GEOMETRIC) && iparam == 0) return -1;
    for (size_t i = 0; i < n; ++i) {
        const uint64_t gi = index_base + i;
        uint64_t lo = 0, hi = 0;
        switch (gen) {
        case ORC_GEN_UNIFORM:
            lo = orc_rand64(seed, gi);
            hi = orc_rand64(seed ^ 0xA5A5A5A5A5A5A5A5ull, gi);
            break;
        case ORC_GEN_ZIPF: { /* x = floor(2^(u * bits)) - 1: the continuous inverse of H(x) = ln x */
            const uint64_t r = orc_rand64(seed, gi);
            const uint64_t t = (r >> 32) * (uint64_t)(bits >= 64 ? 64u : bits);
            lo = orc_exp2_floor((uint32_t)(t >> 32), (uint32_t)t) - 1;
            break;
        }
        case ORC_GEN_STEP: { /* distr.rs:85-93: s = MAX / (n + 1); values s, 2s, .., n s */
            const uint64_t maxv = bits >= 64 ? ~0ull : ((1ull << bits) - 1);
            const uint64_t s = maxv / (iparam + 1);
            lo = s * (1 + orc_rand64(seed, gi) % iparam);
            break;
        }
        case ORC_GEN_SORTED: lo = gi; break;
        case ORC_GEN_REVERSED: lo = index_base + n - 1 - gi; break;
        case ORC_GEN_GEOMETRIC: { /* failures before the first success, by inversion: floor(log2 U / log2(1 - p)) */
            uint64_t w = orc_rand64(seed, gi);
            if (w == 0) w = 1;
            lo = orc_neg_log2(w) / iparam;
            break;
        }
        default: lo = iparam; break;
        }
        uint8_t *e = p + i * L->elem_bytes;
        uint32_t pb = 0;
        for (uint32_t b = 0; b < L->elem_bytes; ++b) {
            if (b >= L->key_offset && b < L->key_offset + L->key_bytes) {
                const uint32_t kb = b - L->key_offset;
                e[b] = (uint8_t)((kb < 8 ? lo >> (8 * kb) : hi >> (8 * (kb - 8))) & 0xFF);
            } else {
                e[b] = (pb < 8 && !payload_zero) ? (uint8_t)((gi >> (8 * pb)) & 0xFF) : 0;
                ++pb;
            }
        }
    }
    return 0;
}

/* ---- the reference's optimisation ladder (mod.rs:40-59,178-571) ------------------------------
 * the experimental variants the reference keeps beside its production sort, restated
 * as CPU ablation baselines (tools/cpu_ladder.py times them).  All give the same bytes as
 * orc_radix_sort -- they differ in threading, in how the scratch buffer is made and in write
 * buffering:
 *   0  radix_sort0  single thread, zero-initialised scratch                     mod.rs:183-212
 *   1  radix_sort1  one thread per digit counts the whole array (a digit's histogram does not
 *                   depend on the order), then serial scatter passes           mod.rs:215-259
 *   2  radix_sort2  chunk per thread: parallel count, digit-major/chunk-minor scan, parallel
 *                   unbuffered scatter; zero-initialised scratch               mod.rs:262-335
 *   3  radix_sort3  = 2 with an uninitialised scratch touched one byte per page mod.rs:338-424
 *   4  radix_sort4  = 3 on a work pool with two chunks per worker (rayon)      mod.rs:427-492
 *   5  radix_sort5  = 3 + 96-element write buffers: the production sort        mod.rs:495-570
 *   counting_sort   u8 only: count, scan, place, copy back                     mod.rs:40-59 */
static inline __attribute__((always_inline)) void orc_scatter_plain_body(orc_job *j, const size_t s) { /* mod.rs:318-327 */
    for (size_t i = j->begin; i < j->end; ++i) {
        const uint8_t *e = j->src + i * s;
        uint32_t v = orc_digit(e, j->L, j->digit);
        memcpy(j->dst + j->hist[v] * s, e, s);
        j->hist[v]++;
    }
}
static void *orc_scatter_plain_worker(void *arg) {
    orc_job *j = (orc_job *)arg;
    ORC_BY_SIZE(orc_scatter_plain_body, j)
    return NULL;
}

typedef struct { /* a pool worker: pulls job indices until none are left (variant 4) */
    orc_job *jobs;
    size_t njobs;
    size_t *next; /* shared cursor */
    void *(*fn)(void *);
} orc_pool_arg;
static void *orc_pool_worker(void *arg) {
    orc_pool_arg *p = (orc_pool_arg *)arg;
    for (;;) {
        size_t k = __atomic_fetch_add(p->next, 1, __ATOMIC_RELAXED);
        if (k >= p->njobs) return NULL;
        p->fn(&p->jobs[k]);
    }
}
/* runs fn over the jobs: one thread per job, or `pool` (> 0) threads pulling jobs */
static int orc_run(orc_job *jobs, size_t njobs, void *(*fn)(void *), pthread_t *tids, int pool) {
    if (njobs == 1) {
        fn(&jobs[0]);
        return 0;
    }
    if (pool > 0) {
        size_t next = 0;
        orc_pool_arg pa = {jobs, njobs, &next, fn};
        int started = 0;
        for (int t = 0; t < pool; ++t, ++started)
            if (pthread_create(&tids[t], NULL, orc_pool_worker, &pa)) break;
        for (int t = 0; t < started; ++t) pthread_join(tids[t], NULL);
        return started == pool ? 0 : -1;
    }
    for (size_t c = 0; c < njobs; ++c)
        if (pthread_create(&tids[c], NULL, fn, &jobs[c])) {
            for (size_t k = 0; k < c; ++k) pthread_join(tids[k], NULL);
            return -1;
        }
    for (size_t c = 0; c < njobs; ++c) pthread_join(tids[c], NULL);
    return 0;
}

int orc_radix_sort_variant(void *data, size_t n, const orc_layout *L, int threads, int variant) {
    const size_t s = L->elem_bytes;
    if (n == 0) return 0;
    if (variant == 0) return orc_radix_sort0(data, n, L);
    if (variant == 5) return orc_radix_sort(data, n, L, threads);
    if (variant < 1 || variant > 4) return -1;
    if (threads < 1) threads = 1;
    const uint32_t D = L->key_bytes;
    uint8_t *self = (uint8_t *)data;
    int rc = 0;
    if (variant == 1) {
        /* one job per digit over the whole (unsorted) array: counts, then each digit's exclusive scan */
        uint8_t *temp = (uint8_t *)calloc(n, s); /* vec![T::default(); len] */
        orc_job *jobs = (orc_job *)calloc(D, sizeof(orc_job));
        pthread_t *tids = (pthread_t *)calloc(D, sizeof(pthread_t));
        if (!temp || !jobs || !tids) rc = -1;
        if (!rc) {
            for (uint32_t d = 0; d < D; ++d) {
                jobs[d].src = self;
                jobs[d].begin = 0;
                jobs[d].end = n;
                jobs[d].L = L;
                jobs[d].digit = d;
            }
            rc = orc_run(jobs, D, orc_count_worker, tids, 0);
        }
        for (uint32_t d = 0; d < D && !rc; ++d) {
            size_t start = 0;
            for (int v = 0; v < 256; ++v) {
                size_t c = jobs[d].hist[v];
                jobs[d].hist[v] = start;
                start += c;
            }
            jobs[d].src = (d % 2 == 0) ? self : temp;
            jobs[d].dst = (d % 2 == 0) ? temp : self;
            orc_scatter_plain_worker(&jobs[d]); /* serial scatter (mod.rs:246-254) */
        }
        if (!rc && D % 2 == 1) memcpy(self, temp, n * s);
        free(tids);
        free(jobs);
        free(temp);
        return rc;
    }
    /* variants 2-4: chunked */
    const size_t parts = variant == 4 ? (size_t)threads * 2 : (size_t)threads; /* CHUNK_MULTIPLIER, mod.rs:430 */
    const size_t per = (n + parts - 1) / parts;
    const size_t chunks = (n + per - 1) / per;
    uint8_t *temp = variant == 2 ? (uint8_t *)calloc(n, s) : (uint8_t *)malloc(n * s);
    orc_job *jobs = (orc_job *)calloc(chunks, sizeof(orc_job));
    pthread_t *tids = (pthread_t *)calloc(chunks > (size_t)threads ? chunks : (size_t)threads, sizeof(pthread_t));
    if (!temp || !jobs || !tids) {
        rc = -1;
        goto out;
    }
    if (variant != 2)
        for (size_t b = 0; b < n * s; b += ORC_PAGE_SIZE) temp[b] = 0; /* page touch (mod.rs:346-357) */
    for (size_t c = 0; c < chunks; ++c) {
        jobs[c].begin = c * per;
        jobs[c].end = (c + 1) * per < n ? (c + 1) * per : n;
        jobs[c].L = L;
    }
    for (uint32_t d = 0; d < D && !rc; ++d) {
        for (size_t c = 0; c < chunks; ++c) {
            jobs[c].src = (d % 2 == 0) ? self : temp;
            jobs[c].dst = (d % 2 == 0) ? temp : self;
            jobs[c].digit = d;
        }
        rc = orc_run(jobs, chunks, orc_count_worker, tids, variant == 4 ? threads : 0);
        if (rc) break;
        size_t prefix = 0;
        for (int v = 0; v < 256; ++v)
            for (size_t c = 0; c < chunks; ++c) {
                size_t cnt = jobs[c].hist[v];
                jobs[c].hist[v] = prefix;
                prefix += cnt;
            }
        rc = orc_run(jobs, chunks, orc_scatter_plain_worker, tids, variant == 4 ? threads : 0);
    }
    if (!rc && D % 2 == 1) memcpy(self, temp, n * s);
out:
    free(tids);
    free(jobs);
    free(temp);
    return rc;
}

/* mod.rs:40-59: counting sort of bytes */
int orc_counting_sort(uint8_t *data, size_t n) {
    uint8_t *temp = (uint8_t *)calloc(n ? n : 1, 1);
    if (!temp) return -1;
    size_t hist[256] = {0};
    for (size_t i = 0; i < n; ++i) hist[data[i]]++;
    size_t start = 0;
    for (int v = 0; v < 256; ++v) {
        size_t c = hist[v];
        hist[v] = start;
        start += c;
    }
    for (size_t i = 0; i < n; ++i) temp[hist[data[i]]++] = data[i];
    memcpy(data, temp, n);
    free(temp);
    return 0;
}
