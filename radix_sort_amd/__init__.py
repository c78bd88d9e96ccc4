"""MI355X-native LSD radix sort: drop-in for the hot path of jgrodzki/radix_sort.

`radix_sort(x)` mirrors `<[T]>::radix_sort(&mut self)` (reference
src/radix_sort/mod.rs:18-20,61-176); `RadixDigits` mirrors the key trait
(src/radix_sort/radix_digits.rs:1-5).  All compute runs in the HIP library
librsx.so behind the C-ABI of include/rsx.h; there is no CPU fallback.
"""
from .api import (PRIMITIVES, Context, RadixDigits, default_context, digits_of, radix_sort, radix_sort_sharded,
                  tuple_of)
from ._lib import (GEN_CONSTANT, GEN_GEOMETRIC, GEN_PAYLOAD_ZERO, GEN_REVERSED, GEN_SORTED, GEN_STEP, GEN_UNIFORM,
                   GEN_ZIPF, INFO_L2_LOCAL, INFO_LAST_PASSES, INFO_RANK_ATOMIC, KEY_FLOAT, KEY_SIGNED, KEY_UNSIGNED, OPT_BYTE_COUNTING,
                   OPT_HOT_LANES, OPT_MAX_REGIONS, OPT_MID_SORT, OPT_WIDE_SORT, OPT_BUCKET_SKIP, OPT_BUCKET_GROUP, OPT_RANK_CHECK, OPT_RANKING, OPT_SMALL_SORT, OPT_STATUS_SCOPE, OPT_TILE_SCHEDULE, OPT_VERBOSE,
                   OPT_XCD_MAJOR, SHARD_EXCHANGE_FIRST, SHARD_SORT_FIRST, Layout, RsxError)

__all__ = ["radix_sort", "radix_sort_sharded", "RadixDigits", "PRIMITIVES", "tuple_of", "digits_of", "Context", "default_context",
           "Layout", "RsxError", "KEY_UNSIGNED", "KEY_SIGNED", "KEY_FLOAT", "GEN_UNIFORM", "GEN_ZIPF", "GEN_STEP",
           "GEN_SORTED", "GEN_REVERSED", "GEN_CONSTANT", "GEN_GEOMETRIC", "GEN_PAYLOAD_ZERO", "OPT_TILE_SCHEDULE",
           "OPT_RANKING", "OPT_STATUS_SCOPE", "OPT_XCD_MAJOR", "OPT_BYTE_COUNTING", "OPT_MAX_REGIONS", "OPT_HOT_LANES",
           "OPT_VERBOSE", "OPT_RANK_CHECK", "OPT_SMALL_SORT", "OPT_MID_SORT", "OPT_WIDE_SORT", "OPT_BUCKET_SKIP", "OPT_BUCKET_GROUP", "INFO_RANK_ATOMIC", "INFO_L2_LOCAL", "INFO_LAST_PASSES", "SHARD_EXCHANGE_FIRST", "SHARD_SORT_FIRST"]
