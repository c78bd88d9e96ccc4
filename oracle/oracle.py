"""ctypes front-end to the CPU oracle (oracle/rsx_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() as the checker.  The product package (radix_sort_amd/)
never imports this module.

Also holds `numpy_stable_sort`, an independent statement of the property the
reference's own tests pin (src/radix_sort/tests.rs:7-23 integers == slice::sort,
:133-173 floats == sort_by(total_cmp) bitwise, :175-187 tuples == stable
sort_by_key(.0)): a stable sort by the mapped key.  The C oracle is pinned
against it in tests/test_oracle.py.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

UNSIGNED, SIGNED, FLOAT = 0, 1, 2


class Layout(ctypes.Structure):
    _fields_ = [
        ("elem_bytes", ctypes.c_uint32),
        ("key_offset", ctypes.c_uint32),
        ("key_bytes", ctypes.c_uint32),
        ("key_kind", ctypes.c_uint32),
    ]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liborc.so")
    src = os.path.join(_HERE, "rsx_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liborc.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liborc.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        vp, sz, lp = ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(Layout)
        L.orc_radix_sort0.argtypes = [vp, sz, lp]
        L.orc_radix_sort0.restype = ctypes.c_int
        L.orc_radix_sort.argtypes = [vp, sz, lp, ctypes.c_int]
        L.orc_radix_sort.restype = ctypes.c_int
        L.orc_partition_pass.argtypes = [vp, vp, sz, lp, ctypes.c_uint32, vp]
        L.orc_partition_pass.restype = ctypes.c_int
        L.orc_map_keys.argtypes = [vp, sz, lp, vp]
        L.orc_map_keys.restype = None
        L.orc_get_digit.argtypes = [vp, lp, ctypes.c_uint32]
        L.orc_get_digit.restype = ctypes.c_uint32
        L.orc_rand64.argtypes = [ctypes.c_uint64, ctypes.c_uint64]
        L.orc_rand64.restype = ctypes.c_uint64
        L.orc_generate.argtypes = [vp, sz, lp, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int]
        L.orc_generate.restype = ctypes.c_int
        L.orc_radix_sort_variant.argtypes = [vp, sz, lp, ctypes.c_int, ctypes.c_int]
        L.orc_radix_sort_variant.restype = ctypes.c_int
        L.orc_counting_sort.argtypes = [vp, sz]
        L.orc_counting_sort.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _as_bytes(a: np.ndarray, elem_bytes: int) -> np.ndarray:
    a = np.ascontiguousarray(a)
    raw = a.view(np.uint8).reshape(-1)
    assert raw.size % elem_bytes == 0
    return raw


def sort0(raw: np.ndarray, layout: Layout) -> np.ndarray:
    """Single-thread oracle (mod.rs:183-212). `raw`: uint8 array of n*elem_bytes; returns a sorted copy."""
    out = np.array(raw, dtype=np.uint8, copy=True)
    n = out.size // layout.elem_bytes
    rc = lib().orc_radix_sort0(out.ctypes.data, n, ctypes.byref(layout))
    assert rc == 0
    return out


def sort_parallel(raw: np.ndarray, layout: Layout, threads: int) -> np.ndarray:
    """Thread-parallel oracle (mod.rs:61-176)."""
    out = np.array(raw, dtype=np.uint8, copy=True)
    n = out.size // layout.elem_bytes
    rc = lib().orc_radix_sort(out.ctypes.data, n, ctypes.byref(layout), threads)
    assert rc == 0
    return out


def sort_parallel_inplace(raw: np.ndarray, layout: Layout, threads: int) -> None:
    n = raw.size // layout.elem_bytes
    rc = lib().orc_radix_sort(raw.ctypes.data, n, ctypes.byref(layout), threads)
    assert rc == 0


def sort_variant(raw: np.ndarray, layout: Layout, threads: int, variant: int) -> np.ndarray:
    """The reference's optimisation ladder radix_sort0..5 (mod.rs:178-571): variant 0..5, same output each."""
    out = np.array(raw, dtype=np.uint8, copy=True)
    n = out.size // layout.elem_bytes
    rc = lib().orc_radix_sort_variant(out.ctypes.data, n, ctypes.byref(layout), threads, variant)
    assert rc == 0
    return out


def sort_variant_inplace(raw: np.ndarray, layout: Layout, threads: int, variant: int) -> None:
    n = raw.size // layout.elem_bytes
    rc = lib().orc_radix_sort_variant(raw.ctypes.data, n, ctypes.byref(layout), threads, variant)
    assert rc == 0


def counting_sort(raw: np.ndarray) -> np.ndarray:
    """counting_sort of mod.rs:40-59 (u8 only)."""
    out = np.array(raw, dtype=np.uint8, copy=True)
    assert lib().orc_counting_sort(out.ctypes.data, out.size) == 0
    return out


def partition_pass(raw: np.ndarray, layout: Layout, digit: int):
    """One LSD pass by `digit`; returns (partitioned copy, 256 counts)."""
    out = np.empty_like(raw)
    hist = np.zeros(256, dtype=np.uint64)
    n = raw.size // layout.elem_bytes
    rc = lib().orc_partition_pass(raw.ctypes.data, out.ctypes.data, n, ctypes.byref(layout), digit, hist.ctypes.data)
    assert rc == 0
    return out, hist


GEN_UNIFORM, GEN_ZIPF, GEN_STEP, GEN_SORTED, GEN_REVERSED, GEN_CONSTANT, GEN_GEOMETRIC = range(7)


def generate(n: int, layout: Layout, gen: int, seed: int, param: float = 0.0, index_base: int = 0,
             payload_zero: bool = False) -> np.ndarray:
    """CPU restatement of rsx_generate_device: n elements as raw bytes.  `param` as the library takes it
    (Zipf: exponent, must be 1; step: number of values; constant: the value; geometric: p)."""
    import math
    if gen == GEN_ZIPF:
        assert param == 1.0, "only the exponent-1 Zipf shape is reproducible across host and device"
        ip = 0
    elif gen in (GEN_STEP, GEN_CONSTANT):
        ip = int(param)
    elif gen == GEN_GEOMETRIC:
        c = -math.log2(1.0 - param) * 4294967296.0  # the library's expression (rsx.hip, rsx_generate_device)
        ip = 1 if c < 1.0 else (2 ** 64 - 1 if c >= 18446744073709551615.0 else int(c))
    else:
        ip = 0
    out = np.zeros(n * layout.elem_bytes, dtype=np.uint8)
    rc = lib().orc_generate(out.ctypes.data, n, ctypes.byref(layout), gen, seed & (2 ** 64 - 1), ip, index_base,
                            1 if payload_zero else 0)
    assert rc == 0
    return out


def map_keys(raw: np.ndarray, layout: Layout) -> np.ndarray:
    n = raw.size // layout.elem_bytes
    out = np.empty(n * layout.key_bytes, dtype=np.uint8)
    lib().orc_map_keys(raw.ctypes.data, n, ctypes.byref(layout), out.ctypes.data)
    return out.reshape(n, layout.key_bytes)


# ---------------------------------------------------------------------------
# Independent numpy statement of the reference tests' acceptance property.
# ---------------------------------------------------------------------------
def numpy_mapped_key_columns(raw: np.ndarray, layout: Layout) -> np.ndarray:
    """Mapped key as (n, key_bytes) uint8 little-endian, computed with numpy only
    (radix_digits.rs:7-124 restated on byte columns, independently of the C code)."""
    n = raw.size // layout.elem_bytes
    e = raw.reshape(n, layout.elem_bytes)
    k = e[:, layout.key_offset : layout.key_offset + layout.key_bytes].copy()
    top = layout.key_bytes - 1
    if layout.key_kind == SIGNED:
        k[:, top] ^= 0x80
    elif layout.key_kind == FLOAT:
        neg = (k[:, top] & 0x80) != 0
        k[neg, :] ^= 0xFF
        k[~neg, top] ^= 0x80
    return k


def numpy_stable_sort(raw: np.ndarray, layout: Layout) -> np.ndarray:
    """Stable sort of the elements by mapped key (np.lexsort is stable; last key is primary)."""
    n = raw.size // layout.elem_bytes
    if n == 0:
        return raw.copy()
    k = numpy_mapped_key_columns(raw, layout)
    if layout.key_bytes <= 8:
        pad = np.zeros((n, 8), dtype=np.uint8)
        pad[:, : layout.key_bytes] = k
        order = np.argsort(pad.view("<u8").reshape(n), kind="stable")
    else:
        lo = np.ascontiguousarray(k[:, :8]).view("<u8").reshape(n)
        hi = np.ascontiguousarray(k[:, 8:16]).view("<u8").reshape(n)
        order = np.lexsort((lo, hi))
    return raw.reshape(n, layout.elem_bytes)[order].reshape(-1).copy()
