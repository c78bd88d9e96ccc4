/*
 * rsx.h -- C-ABI of the MI355X-native LSD radix sort (librsx.so).
 *
 * This is the drop-in boundary for the hot path of jgrodzki/radix_sort:
 *     impl<T: RadixDigits> RadixSort<T> for [T] { fn radix_sort(&mut self) }
 *         reference src/radix_sort/mod.rs:18-20,61-176
 *     trait RadixDigits { const NUMBER_OF_DIGITS: u8; fn get_digit(&self, u8) -> u8 }
 *         reference src/radix_sort/radix_digits.rs:1-5 (+ impls :7-136)
 * The reference has no FFI layer of its own (pure Rust, bin crate); these are
 * the entry points a Rust `extern "C"` block would bind so that the body of
 * `radix_sort()` becomes one call (binding shown in INTEGRATION.md).
 *
 * Plain pointers and sizes only; no C++/torch types.  Every function returns
 * an `int` status (0 = RSX_OK, negative = error) and never unwinds.
 */
#ifndef RSX_H
#define RSX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSX_VERSION 200 /* 0.2.0 */

/* status codes */
enum {
    RSX_OK = 0,
    RSX_ERR_ARG = -1,         /* bad pointer / size / layout */
    RSX_ERR_UNSUPPORTED = -2, /* element size or key width without a device kernel */
    RSX_ERR_HIP = -3,         /* a HIP runtime call failed; see rsx_last_error */
    RSX_ERR_NOMEM = -4,       /* device workspace allocation failed */
    RSX_ERR_NODEVICE = -5,    /* no gfx950-class device / wrong device */
    RSX_ERR_WORKSPACE = -6,   /* the call would have to allocate while its stream is being captured:
                                 rsx_ctx_reserve first */
    RSX_ERR_INTERNAL = -7     /* a bounded device-side wait gave up (protocol error); results invalid */
};

/* How a key is mapped to its order-preserving unsigned form before digits are
 * taken; restates radix_digits.rs. */
enum {
    RSX_KEY_UNSIGNED = 0, /* u8,u16,u32,u64,u128,usize      radix_digits.rs:7-53   */
    RSX_KEY_SIGNED = 1,   /* i8..i128,isize: x ^ MIN        radix_digits.rs:55-101 */
    RSX_KEY_FLOAT = 2     /* f32,f64: b ^= (b>>31)|MIN      radix_digits.rs:103-124 */
};

/* Element descriptor: what `T: RadixDigits` means to the device.
 *   elem_bytes  size_of::<T>()            (1,2,4,8,12,16,24,32 have kernels)
 *   key_offset  byte offset of the key inside the element (0 for primitives;
 *               offset_of!((K,U), 0) for tuples, radix_digits.rs:126-136)
 *   key_bytes   1,2,4,8,16 == T::NUMBER_OF_DIGITS (8-bit digits, LSB first)
 *   key_kind    RSX_KEY_*
 * Elements are moved bitwise (mod.rs:133-140 uses copy_nonoverlapping), so any
 * payload -- padding included -- is carried unchanged. */
typedef struct rsx_layout {
    uint32_t elem_bytes;
    uint32_t key_offset;
    uint32_t key_bytes;
    uint32_t key_kind;
} rsx_layout;

/* Owns the device workspace of its sorts.  Any number of threads may call into one context (calls
 * are serialised by a mutex) and its calls may name different streams: work enqueued on a stream
 * other than the context's previous one first waits, on the device, for that previous work (the
 * workspace belongs to one sort at a time).  Use one context per stream to run sorts concurrently. */
typedef struct rsx_ctx rsx_ctx;

/* -- context ------------------------------------------------------------- */
/* Binds a context to HIP device `device` (-1 = current device). */
int rsx_ctx_create(int device, rsx_ctx **out);
int rsx_ctx_destroy(rsx_ctx *ctx);
/* Pre-allocates the internal workspace for sorts of up to `n` elements of
 * `layout` so that rsx_sort_device performs no allocation (stream-capture
 * safe).  Replaces the reference's per-call temp-buffer allocation + page
 * touch (mod.rs:71-82) for everything except the caller-owned ping-pong
 * buffer. */
int rsx_ctx_reserve(rsx_ctx *ctx, size_t n, const rsx_layout *layout);
/* Synchronises `stream` and reports RSX_ERR_INTERNAL if any kernel of this
 * context flagged a device-side protocol error since the last check (the
 * reference panics on worker failure, mod.rs:106; across a C ABI that becomes
 * a status), then clears the condition.  The error word is host-visible, so a
 * pending error also fails the NEXT rsx_sort_device / rsx_partition_device on
 * the context without any synchronisation; callers of the stream-ordered entry
 * points should still call this at their own sync point -- it is the only way
 * to learn that the sort just enqueued went wrong.  rsx_sort_host and
 * rsx_sort_sharded check by themselves. */
int rsx_ctx_check(rsx_ctx *ctx, void *stream);
/* Alternative code paths of the sweep kernel; every one gives the same bytes (they exist as
 * fall-backs that the library selects itself when a device self-test fails, and are exposed so
 * that callers and tests can force them). */
enum {
    RSX_OPT_TILE_SCHEDULE = 1, /* 0 (default): static tile assignment behind a start-up roll call, tickets if
                                  it fails; 1: ticketed tiles always */
    RSX_OPT_RANKING = 2,       /* 0 (default): ranks from returned LDS atomics where the device's ordering
                                  self-test passed and the tile is not skewed, wave ballots otherwise;
                                  1: ballots only; 2: LDS atomics whatever the skew */
    RSX_OPT_STATUS_SCOPE = 3,  /* 0 (default): look-back status words of a verified single-XCD chain stay in
                                  that XCD's L2; 1: agent-scope stores everywhere */
    RSX_OPT_XCD_MAJOR = 4,     /* 1 (default): workgroups numbered XCD-major; 0: by blockIdx */
    RSX_OPT_BYTE_COUNTING = 5, /* 1 (default): u8/i8 arrays, and u16/i16 arrays of at least 2^23 elements, by counting
                                  (the element is its key: the histogram is the sorted array); 0: through the general passes */
    RSX_OPT_MAX_REGIONS = 6,   /* 0 (default: 8 or 16 by element size) .. 32 look-back chains per pass */
    RSX_OPT_HOT_LANES = 7,     /* 2..65 (default 16): lanes sharing a digit that mark a tile as skewed */
    RSX_OPT_VERBOSE = 8,       /* 1: launch geometry and self-test verdicts on stderr (also env RSX_VERBOSE=1) */
    RSX_OPT_RANK_CHECK = 9,    /* 1: in every tile, one round of LDS-atomic ranks is cross-checked against the
                                  ballot-derived ranks (the property rsx_lds_order_kernel tests on an idle device,
                                  here under the real sweeps' LDS contention); a mismatch makes rsx_ctx_check fail */
    RSX_OPT_SMALL_SORT = 10,   /* 1 (default): arrays of at most one tile (14336 4-byte, 7168 8-byte, 2560 16-byte
                                  elements ...) are sorted by ONE launch of one workgroup; 0: by the general path */
    RSX_OPT_MID_SORT = 11      /* middle-size arrays (up to 2^22 4-byte, 2^21 8-byte, 2^20 16-byte elements): when their most
                                  significant digit spreads them over its 256 buckets, one sweep makes the buckets and one
                                  workgroup per bucket sorts it in LDS (two trips through memory instead of D).  1 (default):
                                  the host forecasts from what the context's previous middle-size sort reported; an input
                                  that is skewed after all is still sorted correctly (an oversized bucket goes through
                                  memory), then the context keeps to LSD passes for its next sorts.  0: LSD passes always,
                                  top digit not even counted.  2: always split.  3: always LSD passes. */,
    RSX_OPT_WIDE_SORT = 12,    /* large arrays of 8-byte (and wider) keys: count the top 16 bits of the key, two sweeps for
                                  those two digits, then every 16-bit bucket sorted by its remaining digits in LDS.
                                  0: never; 1 (default): by key width and size (8- and 16-byte keys above the middle sizes, 4-byte keys in
                                  8-byte elements from 1 GiB, in wider ones from 2 GiB), when the count says every bucket
                                  fits; 2: always (any array of 65536+ such elements, buckets that do not fit go through
                                  memory); 3: as 1 without the size floor (above the middle sizes) */
    RSX_OPT_BUCKET_SKIP = 13,  /* the LDS passes of the bucket kernels: 1 (default) start at the digit that leaves them the
                                  bits an array of that size is, as a rule, told apart by (2 log2 m - 6 of its variable
                                  bits: three passes for a 16-bit bucket of 2^30 u64 keys) and put right the neighbours
                                  that still agree, by the digits skipped; 0: every pass */
    RSX_OPT_BUCKET_GROUP = 14  /* the hybrid on arrays whose 16-bit buckets are small (8-byte and wider keys): 1 (default)
                                  a workgroup sorts a group of consecutive buckets as one array; 0: bucket by bucket */
};
int rsx_ctx_set_option(rsx_ctx *ctx, int option, uint64_t value);
enum {
    RSX_INFO_RANK_ATOMIC = 1, /* 1 if the LDS atomic ordering self-test passed on this device */
    RSX_INFO_L2_LOCAL = 2,    /* 1 if the same-XCD hand-off self-test passed on this device */
    RSX_INFO_NUM_CU = 3,
    RSX_INFO_DEVICE = 4,
    RSX_INFO_LAST_PASSES = 5  /* which tile schedule the passes of the context's LAST sort ran with (waits for it):
                                 bits 0-7 sweep passes launched (0: one-launch or counting path), bits 8-15 of them
                                 with static tiles (the roll call succeeded), bits 16-23 of them with the XCD
                                 placement verified (status words of single-XCD chains stay in L2), bits 24-27 the
                                 path: 0 general passes, 1 one-launch sort of at most one tile, 2 middle-size bucket
                                 split, 3 one-byte counting, 4 two-byte counting, 5 wide-key hybrid (two sweeps + the
                                 16-bit buckets in LDS; a hybrid the device refused reports 0 and its D passes) */
};
int rsx_ctx_get_info(rsx_ctx *ctx, int what, uint64_t *out);
/* Per-launch timing with HIP events on the launch stream (measurement only).
 * rsx_ctx_profile(ctx, 1) clears the counters and makes every later kernel
 * launch of this context record a start/stop event pair around itself;
 * rsx_ctx_profile(ctx, 0) stops recording.  rsx_ctx_profile_read waits for the
 * recorded events and returns, per kernel kind, the summed duration in
 * milliseconds and the number of launches (arrays of RSX_PROF_KINDS). */
enum { RSX_PROF_HIST = 0, RSX_PROF_SCAN = 1, RSX_PROF_SWEEP = 2, RSX_PROF_OTHER = 3, RSX_PROF_KINDS = 4 };
int rsx_ctx_profile(rsx_ctx *ctx, int enable);
int rsx_ctx_profile_read(rsx_ctx *ctx, double *ms, uint64_t *launches);
/* Last error text for this context (never NULL). */
const char *rsx_last_error(const rsx_ctx *ctx);
const char *rsx_strerror(int status);
int rsx_version(void);

/* -- the sort ------------------------------------------------------------ */
/* In-place ascending stable sort of `n` elements resident in device memory:
 * the device-side body of `<[T]>::radix_sort` (mod.rs:62-175).  `d_tmp` is the
 * ping-pong buffer (the reference's `temp`, mod.rs:71-83), n*elem_bytes bytes,
 * caller-owned.  All work is enqueued on `stream` (a hipStream_t, NULL =
 * default stream); the call does not synchronise the device.  On return the
 * result is (stream-ordered) in `d_data`, as after mod.rs:170-174. */
int rsx_sort_device(rsx_ctx *ctx, void *d_data, void *d_tmp, size_t n, const rsx_layout *layout,
                    void *stream);

/* Literal drop-in for `&mut [T]` in host memory: H2D, rsx_sort_device, D2H,
 * blocking (mod.rs:62 is blocking too).  The copies run as a pipeline over a ring of
 * pinned chunks (the slice itself is pageable).  PCIe-bound; not the measured path. */
int rsx_sort_host(rsx_ctx *ctx, void *data, size_t n, const rsx_layout *layout);

/* -- per-pass building blocks (multi-GPU bucket exchange) ---------------- */
/* 256-bin count of digit `digit` (0 = least significant) over `n` elements:
 * the count phase, mod.rs:90-109, with "chunk" = this device's slice.
 * `d_hist` receives 256 uint64 counts (overwritten). */
int rsx_histogram_device(rsx_ctx *ctx, const void *d_src, size_t n, const rsx_layout *layout,
                         uint32_t digit, uint64_t *d_hist, void *stream);
/* One stable LSD pass by `digit`, d_src -> d_dst (count -> scan -> scatter of
 * mod.rs:90-168 for one current_digit_index).  If `d_hist` is non-NULL it
 * receives the 256 uint64 digit counts of the slice. */
int rsx_partition_device(rsx_ctx *ctx, const void *d_src, void *d_dst, size_t n,
                         const rsx_layout *layout, uint32_t digit, uint64_t *d_hist, void *stream);
/* The same pass over `nsub` (1..16) independent position sub-ranges of the slice, in two steps, so that a
 * multi-GPU driver can put the buckets of sub-range 0 on the links while sub-range 1 is still being scattered:
 * sub-range k = elements [n*k/nsub, n*(k+1)/nsub).
 *   rsx_partition_count_device    counts `digit` over every sub-range (count phase, mod.rs:90-109, chunk ==
 *                                 sub-range): d_hist receives nsub x 256 uint64; the context keeps the count
 *                                 matrices for the scatter calls that follow;
 *   rsx_partition_scatter_device  stable partition of sub-range k by `digit` (prefix + scatter, mod.rs:110-168):
 *                                 its elements land in the same position range of d_dst, grouped by digit.
 * Every scatter call of a count call must name the same src, n, layout, digit and nsub. */
int rsx_partition_count_device(rsx_ctx *ctx, const void *d_src, size_t n, const rsx_layout *layout, uint32_t digit,
                               uint32_t nsub, uint64_t *d_hist, void *stream);
int rsx_partition_scatter_device(rsx_ctx *ctx, const void *d_src, void *d_dst, size_t n, const rsx_layout *layout,
                                 uint32_t digit, uint32_t nsub, uint32_t k, void *stream);

/* Segmented device copy: for i in [0, nseg): copy len[i] ELEMENTS of
 * `elem_bytes` from d_src + src_off[i] to d_dst + dst_off[i] (offsets in
 * elements).  Places the received (digit, source-GPU) runs after the
 * all-to-all -- the "digit-major, chunk-minor" order of mod.rs:110-120 with
 * chunk == GPU.  The three tables are device arrays of nseg uint64. */
int rsx_segmented_copy_device(rsx_ctx *ctx, const void *d_src, void *d_dst, uint32_t elem_bytes,
                              const uint64_t *d_src_off, const uint64_t *d_dst_off,
                              const uint64_t *d_len, uint32_t nseg, void *stream);

/* Lower and upper bounds of `nq` 128-bit mapped-key queries in a slice that is
 * already sorted: d_queries holds nq pairs (low 64 bits, high 64 bits) of the
 * mapped key of radix_digits.rs:7-124 (unsigned order == sort order); d_out
 * receives 2*nq uint64: d_out[i] = elements with key < query i, d_out[nq + i] =
 * elements with key <= query i.  The splitter search of a multi-GPU sort asks
 * these of every locally sorted slice (mod.rs:110-120's cursors, by search
 * instead of by scan). */
int rsx_bounds_device(rsx_ctx *ctx, const void *d_sorted, size_t n, const rsx_layout *layout,
                      const uint64_t *d_queries, uint32_t nq, uint64_t *d_out, void *stream);
/* The same with a range per query: query i is answered inside elements [d_ranges[2i], d_ranges[2i+1]) of
 * `d_data` (a range sorted by mapped key; different queries may name different ranges), counts relative
 * to the range's start.  One call then serves all boundaries of the exchange-first schedule, whose
 * sorted pieces are the top-digit buckets the boundaries fall into. */
int rsx_bounds_ranges_device(rsx_ctx *ctx, const void *d_data, size_t n, const rsx_layout *layout,
                             const uint64_t *d_queries, const uint64_t *d_ranges, uint32_t nq, uint64_t *d_out,
                             void *stream);

/* One digit of the splitter search with the key prefix kept on the device (no host round trip per digit): for
 * boundary b (0 .. nb-1), d_ranges[2b], d_ranges[2b+1] name a range of `d_data` sorted by mapped key and
 * d_prefix[2b], d_prefix[2b+1] hold the low / high 64 bits of the mapped key with the digits above `digit` fixed and
 * the rest zero.  rsx_splitter_count_device writes d_less[b*256 + j] = elements of range b with key below
 * prefix_b | j << 8*digit; the caller sums d_less over the ranks (an all-reduce on the device); then
 * rsx_splitter_pick_device ORs into prefix_b the largest j whose summed count does not exceed d_rank[b].  After digit
 * 0, rsx_bounds_ranges_device with d_queries = d_prefix gives the final (less, less-or-equal) counts. */
int rsx_splitter_count_device(rsx_ctx *ctx, const void *d_data, size_t n, const rsx_layout *layout,
                              const uint64_t *d_ranges, const uint64_t *d_prefix, uint32_t nb, uint32_t digit,
                              uint64_t *d_less, void *stream);
int rsx_splitter_pick_device(rsx_ctx *ctx, const uint64_t *d_total, const uint64_t *d_rank, uint64_t *d_prefix,
                             uint32_t nb, uint32_t digit, void *stream);

/* -- multi-GPU from one process ------------------------------------------- */
/* Sorts the concatenation slice 0 | slice 1 | ... | slice ndev-1 as ONE array,
 * stably and in place: slice g keeps its length n_per_dev[g] and ends up
 * holding elements [sum(n_per_dev[..g]), +n_per_dev[g]) of the sorted whole --
 * bit-identical to rsx_sort_device on the concatenation.  "Chunk per thread"
 * of mod.rs:66-70,90-168 becomes "slice per GPU".  ctxs[g] is bound to the
 * device that holds d_slices[g] and d_tmps[g] (scratch of the same size as the
 * slice); one context per slice, several may share a device.  Blocking; uses
 * one private stream per slice; data crosses devices once (peer copies over
 * xGMI).  Errors are reported on ctxs[0]. */
int rsx_sort_sharded(rsx_ctx *const *ctxs, uint32_t ndev, void *const *d_slices, void *const *d_tmps,
                     const size_t *n_per_dev, const rsx_layout *layout);
/* The same with the schedule named.  Both move every element across devices once and give the
 * same bytes:
 *   RSX_SHARD_EXCHANGE_FIRST (what rsx_sort_sharded runs): one stable partition pass by the most
 *     significant digit per slice, the G x 256 counts laid out globally (mod.rs:110-120 with chunk ==
 *     slice), boundary buckets sorted locally and cut exactly, exchange, ONE local sort;
 *   RSX_SHARD_SORT_FIRST: local sort, exact splitters by search in the sorted slices, exchange,
 *     second local sort. */
enum { RSX_SHARD_EXCHANGE_FIRST = 0, RSX_SHARD_SORT_FIRST = 1 };
int rsx_sort_sharded_ex(rsx_ctx *const *ctxs, uint32_t ndev, void *const *d_slices, void *const *d_tmps,
                        const size_t *n_per_dev, const rsx_layout *layout, int schedule);

/* -- harness helpers (input generation / verification on device) --------- */
enum {
    RSX_GEN_UNIFORM = 0, /* key = splitmix64(seed, i) truncated        (distr.rs:40-52 KeyUniform shape) */
    RSX_GEN_ZIPF = 1,    /* key ~ Zipf-shaped over [0, 2^bits - 1), exponent s = param: floor(2^(u bits)) - 1
                            for s = 1                                   (distr.rs:54-76,108-130)         */
    RSX_GEN_STEP = 2,    /* key uniform over `param` equally spaced values (distr.rs:78-106,132-160)    */
    RSX_GEN_SORTED = 3,  /* key = i (already sorted)                                                   */
    RSX_GEN_REVERSED = 4,/* key = n-1-i                                                                */
    RSX_GEN_CONSTANT = 5,/* key = param                                                                */
    RSX_GEN_GEOMETRIC = 6,/* key ~ Geometric(p = param): failures before the first success (distr.rs:3-38 MyExp) */
    RSX_GEN_PAYLOAD_ZERO = 0x100 /* OR into `gen`: payload bytes are 0, the reference's `(key, 0)` pairs
                                    (distr.rs:22-26,42-52), instead of the element's index */
};
/* Fills `n` elements: key field generated as above -- counter-based (splitmix64 of seed and
 * index) and in integer arithmetic throughout, so the same (seed, index) gives the same key on
 * any device and in the CPU restatement (tests/test_generators.py compares them byte for byte);
 * the one exception is RSX_GEN_ZIPF with param != 1, which uses the device's double-precision
 * pow.  The reference draws from rand_distr with an unseeded thread_rng: these are its
 * distributions' shapes, not its streams.  Every payload byte outside the key holds the low
 * bytes of the element's global index `index_base + i` (reveals instability) unless
 * RSX_GEN_PAYLOAD_ZERO is set. */
int rsx_generate_device(rsx_ctx *ctx, void *d_data, size_t n, const rsx_layout *layout, int gen,
                        uint64_t seed, double param, uint64_t index_base, void *stream);
/* Order check + order-independent checksum, on device:
 *   out[0] = number of adjacent pairs (i, i+1) with mapped_key[i] > mapped_key[i+1]
 *   out[1] = sum over elements of hash(element bytes) mod 2^64 (multiset checksum)
 *   out[2] = number of adjacent equal-key pairs whose payload index decreases
 *            (stability violations; meaningful for rsx_generate_device payloads,
 *            elements with payload bytes only)
 * `d_out` is 3 uint64 on the device (overwritten). */
int rsx_verify_device(rsx_ctx *ctx, const void *d_data, size_t n, const rsx_layout *layout,
                      uint64_t *d_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RSX_H */
