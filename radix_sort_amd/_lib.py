"""ctypes binding of include/rsx.h.  Fails loudly when librsx.so is missing:
there is no CPU fallback in the product path."""
from __future__ import annotations

import ctypes
import os

from . import _build

# every symbol include/rsx.h declares
SYMBOLS = [
    "rsx_ctx_create", "rsx_ctx_destroy", "rsx_ctx_reserve", "rsx_ctx_check", "rsx_ctx_set_option", "rsx_ctx_get_info",
    "rsx_ctx_profile", "rsx_ctx_profile_read", "rsx_last_error",
    "rsx_strerror", "rsx_version", "rsx_sort_device", "rsx_sort_host", "rsx_histogram_device",
    "rsx_partition_device", "rsx_partition_count_device", "rsx_partition_scatter_device", "rsx_splitter_count_device",
    "rsx_splitter_pick_device", "rsx_segmented_copy_device", "rsx_bounds_device", "rsx_bounds_ranges_device", "rsx_sort_sharded",
    "rsx_sort_sharded_ex", "rsx_generate_device", "rsx_verify_device",
]

OK, ERR_ARG, ERR_UNSUPPORTED, ERR_HIP, ERR_NOMEM, ERR_NODEVICE, ERR_WORKSPACE, ERR_INTERNAL = 0, -1, -2, -3, -4, -5, -6, -7
KEY_UNSIGNED, KEY_SIGNED, KEY_FLOAT = 0, 1, 2
PROF_HIST, PROF_SCAN, PROF_SWEEP, PROF_OTHER, PROF_KINDS = 0, 1, 2, 3, 4
GEN_UNIFORM, GEN_ZIPF, GEN_STEP, GEN_SORTED, GEN_REVERSED, GEN_CONSTANT, GEN_GEOMETRIC = 0, 1, 2, 3, 4, 5, 6
GEN_PAYLOAD_ZERO = 0x100
(OPT_TILE_SCHEDULE, OPT_RANKING, OPT_STATUS_SCOPE, OPT_XCD_MAJOR, OPT_BYTE_COUNTING, OPT_MAX_REGIONS, OPT_HOT_LANES,
 OPT_VERBOSE, OPT_RANK_CHECK, OPT_SMALL_SORT, OPT_MID_SORT, OPT_WIDE_SORT, OPT_BUCKET_SKIP, OPT_BUCKET_GROUP) = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14
INFO_RANK_ATOMIC, INFO_L2_LOCAL, INFO_NUM_CU, INFO_DEVICE, INFO_LAST_PASSES = 1, 2, 3, 4, 5
SHARD_EXCHANGE_FIRST, SHARD_SORT_FIRST = 0, 1


class Layout(ctypes.Structure):
    """struct rsx_layout (include/rsx.h)."""
    _fields_ = [
        ("elem_bytes", ctypes.c_uint32),
        ("key_offset", ctypes.c_uint32),
        ("key_bytes", ctypes.c_uint32),
        ("key_kind", ctypes.c_uint32),
    ]

    def __repr__(self):
        return f"Layout(elem_bytes={self.elem_bytes}, key_offset={self.key_offset}, key_bytes={self.key_bytes}, key_kind={self.key_kind})"


class RsxError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"rsx status {status}: {msg}")
        self.status = status


_LIB = None


def lib_path() -> str:
    # RSX_LIBRARY: another build of the same library (A/B timing of build-time knobs)
    return os.environ.get("RSX_LIBRARY") or _build.LIB


def load():
    """Loads librsx.so (must have been built by _build.build / __graft_entry__.build)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # A process must hold ONE HIP runtime.  torch ships its own libamdhip64 and librsx.so links the
    # system one under the same soname: whichever loads first serves both, but torch cannot see the GPU
    # when it comes second.  So where torch exists (tensors, streams), it is imported first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -m radix_sort_amd._build` "
            "(needs hipcc); there is no CPU fallback for the sort path")
    L = ctypes.CDLL(path)
    vp, sz, lp = ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(Layout)
    u32, u64, i = ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int
    L.rsx_ctx_create.argtypes = [i, ctypes.POINTER(vp)]
    L.rsx_ctx_destroy.argtypes = [vp]
    L.rsx_ctx_reserve.argtypes = [vp, sz, lp]
    L.rsx_ctx_check.argtypes = [vp, vp]
    L.rsx_ctx_set_option.argtypes = [vp, i, u64]
    L.rsx_ctx_get_info.argtypes = [vp, i, ctypes.POINTER(ctypes.c_uint64)]
    L.rsx_ctx_profile.argtypes = [vp, i]
    L.rsx_ctx_profile_read.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64)]
    L.rsx_last_error.argtypes = [vp]
    L.rsx_last_error.restype = ctypes.c_char_p
    L.rsx_strerror.argtypes = [i]
    L.rsx_strerror.restype = ctypes.c_char_p
    L.rsx_version.argtypes = []
    L.rsx_sort_device.argtypes = [vp, vp, vp, sz, lp, vp]
    L.rsx_sort_host.argtypes = [vp, vp, sz, lp]
    L.rsx_histogram_device.argtypes = [vp, vp, sz, lp, u32, vp, vp]
    L.rsx_partition_device.argtypes = [vp, vp, vp, sz, lp, u32, vp, vp]
    L.rsx_partition_count_device.argtypes = [vp, vp, sz, lp, u32, u32, vp, vp]
    L.rsx_partition_scatter_device.argtypes = [vp, vp, vp, sz, lp, u32, u32, u32, vp]
    L.rsx_splitter_count_device.argtypes = [vp, vp, sz, lp, vp, vp, u32, u32, vp, vp]
    L.rsx_splitter_pick_device.argtypes = [vp, vp, vp, vp, u32, u32, vp]
    L.rsx_segmented_copy_device.argtypes = [vp, vp, vp, u32, vp, vp, vp, u32, vp]
    L.rsx_bounds_device.argtypes = [vp, vp, sz, lp, vp, u32, vp, vp]
    L.rsx_bounds_ranges_device.argtypes = [vp, vp, sz, lp, vp, vp, u32, vp, vp]
    L.rsx_sort_sharded.argtypes = [vp, u32, vp, vp, vp, lp]
    L.rsx_sort_sharded_ex.argtypes = [vp, u32, vp, vp, vp, lp, i]
    L.rsx_generate_device.argtypes = [vp, vp, sz, lp, i, u64, ctypes.c_double, u64, vp]
    L.rsx_verify_device.argtypes = [vp, vp, sz, lp, vp, vp]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if name not in ("rsx_last_error", "rsx_strerror"):
            fn.restype = i
    _LIB = L
    return L
