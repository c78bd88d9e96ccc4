"""Python host mirror of the reference's operator interface over the C-ABI.

Reference interface (src/radix_sort/mod.rs:18-20, radix_digits.rs:1-5):

    pub trait RadixDigits { const NUMBER_OF_DIGITS: u8; fn get_digit(&self, index: u8) -> u8; }
    pub trait RadixSort<T: RadixDigits> { fn radix_sort(&mut self); }   // impl for [T]

Here `RadixDigits` is a descriptor (what the Rust shim forwards as `rsx_layout`)
and `radix_sort(x)` sorts `x` in place, ascending, stably, by the mapped key --
same name, same argument meaning, blocking for host arrays like mod.rs:62.
torch is used for device memory and streams only.
"""
from __future__ import annotations

import ctypes
import threading
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import KEY_FLOAT, KEY_SIGNED, KEY_UNSIGNED, Layout, RsxError


@dataclass(frozen=True)
class RadixDigits:
    """Key model of one element type (radix_digits.rs).  NUMBER_OF_DIGITS == key_bytes."""
    elem_bytes: int
    key_offset: int
    key_bytes: int
    key_kind: int

    @property
    def NUMBER_OF_DIGITS(self) -> int:  # noqa: N802 (reference name)
        return self.key_bytes

    def layout(self) -> Layout:
        return Layout(self.elem_bytes, self.key_offset, self.key_bytes, self.key_kind)

    def get_digit(self, element: bytes, index: int) -> int:
        """radix_digits.rs get_digit on the little-endian bytes of one element (host-side helper)."""
        k = element[self.key_offset:self.key_offset + self.key_bytes]
        top = self.key_bytes - 1
        b = k[index]
        if self.key_kind == KEY_SIGNED:
            if index == top:
                b ^= 0x80
        elif self.key_kind == KEY_FLOAT:
            if k[top] & 0x80:
                b ^= 0xFF
            elif index == top:
                b ^= 0x80
        return b


def _prim(name: str) -> RadixDigits:
    kinds = {"u": KEY_UNSIGNED, "i": KEY_SIGNED, "f": KEY_FLOAT}
    bits = int(name[1:])
    return RadixDigits(bits // 8, 0, bits // 8, kinds[name[0]])


#: the reference's built-in impls (radix_digits.rs:7-124); usize/isize are 64-bit
PRIMITIVES = {n: _prim(n) for n in
              ("u8", "u16", "u32", "u64", "u128", "i8", "i16", "i32", "i64", "i128", "f32", "f64")}
PRIMITIVES["usize"] = PRIMITIVES["u64"]
PRIMITIVES["isize"] = PRIMITIVES["i64"]


def tuple_of(key: str, payload_bytes: int, key_offset: Optional[int] = None,
             elem_bytes: Optional[int] = None) -> RadixDigits:
    """`(K, U)` (radix_digits.rs:126-136): key `.0` of primitive `key`, opaque payload.
    Default layout = key first, payload after it, size rounded up to the key alignment
    (what rustc does for (K, U) with size_of::<U>() <= size_of::<K>(); pass explicit
    offsets for anything else -- Rust tuple layout is not ABI-stable)."""
    k = PRIMITIVES[key]
    off = 0 if key_offset is None else key_offset
    if elem_bytes is None:
        al = min(k.key_bytes, 16)
        elem_bytes = -(-(k.key_bytes + payload_bytes) // al) * al
    return RadixDigits(elem_bytes, off, k.key_bytes, k.key_kind)


_NP_KIND = {"u": KEY_UNSIGNED, "i": KEY_SIGNED, "f": KEY_FLOAT}


def digits_of(dtype) -> RadixDigits:
    """RadixDigits of a numpy dtype: primitives, or a structured dtype whose FIRST field is the key."""
    dt = np.dtype(dtype)
    if dt.fields:
        name0 = dt.names[0]
        kdt, koff = dt.fields[name0][0], dt.fields[name0][1]
        if kdt.kind not in _NP_KIND or kdt.itemsize not in (1, 2, 4, 8):
            if kdt.kind == "V" and kdt.itemsize == 16:  # u128 key stored as 16 raw bytes
                return RadixDigits(dt.itemsize, koff, 16, KEY_UNSIGNED)
            raise TypeError(f"unsupported key field dtype {kdt}")
        return RadixDigits(dt.itemsize, koff, kdt.itemsize, _NP_KIND[kdt.kind])
    if dt.kind in _NP_KIND and dt.itemsize in (1, 2, 4, 8):
        if dt.kind == "f" and dt.itemsize not in (4, 8):
            raise TypeError(f"unsupported float width {dt}")
        return RadixDigits(dt.itemsize, 0, dt.itemsize, _NP_KIND[dt.kind])
    raise TypeError(f"no RadixDigits for dtype {dt}; pass digits= explicitly")


class Context:
    """rsx_ctx: owns the device workspace (replaces the per-call temp alloc of mod.rs:71-82)."""

    def __init__(self, device: int = -1):
        self._L = _lib.load()
        h = ctypes.c_void_p()
        rc = self._L.rsx_ctx_create(device, ctypes.byref(h))
        if rc != 0:
            raise RsxError(rc, self._L.rsx_strerror(rc).decode())
        self._h = h

    def _check(self, rc: int):
        if rc != 0:
            raise RsxError(rc, f"{self._L.rsx_strerror(rc).decode()} ({self._L.rsx_last_error(self._h).decode()})")

    def close(self):
        if getattr(self, "_h", None):
            self._L.rsx_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- raw C-ABI calls (pointers are ints) ------------------------------------
    def reserve(self, n: int, d: RadixDigits):
        lay = d.layout()
        self._check(self._L.rsx_ctx_reserve(self._h, n, ctypes.byref(lay)))

    def check(self, stream: int = 0):
        """rsx_ctx_check: synchronises `stream` and raises if a kernel of this context gave up a
        device-side wait (the stream-ordered entry points cannot report that by themselves)."""
        self._check(self._L.rsx_ctx_check(self._h, stream))

    def set_option(self, option: int, value: int):
        """rsx_ctx_set_option (OPT_* in _lib): forces one of the bit-exact alternative kernel paths."""
        self._check(self._L.rsx_ctx_set_option(self._h, option, value))

    def get_info(self, what: int) -> int:
        out = ctypes.c_uint64(0)
        self._check(self._L.rsx_ctx_get_info(self._h, what, ctypes.byref(out)))
        return int(out.value)

    def profile(self, enable: bool):
        """Per-launch HIP-event timing on/off (rsx_ctx_profile); enabling clears the counters."""
        self._check(self._L.rsx_ctx_profile(self._h, 1 if enable else 0))

    def profile_read(self):
        """-> {kind: (total_ms, launches)} for kinds hist/scan/sweep/other."""
        ms = (ctypes.c_double * _lib.PROF_KINDS)()
        cnt = (ctypes.c_uint64 * _lib.PROF_KINDS)()
        self._check(self._L.rsx_ctx_profile_read(self._h, ms, cnt))
        names = ("hist", "scan", "sweep", "other")
        return {names[k]: (ms[k], int(cnt[k])) for k in range(_lib.PROF_KINDS)}

    def sort_device(self, d_data: int, d_tmp: int, n: int, d: RadixDigits, stream: int = 0):
        lay = d.layout()
        self._check(self._L.rsx_sort_device(self._h, d_data, d_tmp, n, ctypes.byref(lay), stream))

    def sort_host(self, ptr: int, n: int, d: RadixDigits):
        lay = d.layout()
        self._check(self._L.rsx_sort_host(self._h, ptr, n, ctypes.byref(lay)))

    def histogram_device(self, d_src: int, n: int, d: RadixDigits, digit: int, d_hist: int, stream: int = 0):
        lay = d.layout()
        self._check(self._L.rsx_histogram_device(self._h, d_src, n, ctypes.byref(lay), digit, d_hist, stream))

    def partition_device(self, d_src: int, d_dst: int, n: int, d: RadixDigits, digit: int, d_hist: int = 0,
                         stream: int = 0):
        lay = d.layout()
        self._check(self._L.rsx_partition_device(self._h, d_src, d_dst, n, ctypes.byref(lay), digit, d_hist, stream))

    def partition_count_device(self, d_src: int, n: int, d: RadixDigits, digit: int, nsub: int, d_hist: int, stream: int = 0):
        lay = d.layout()
        self._check(self._L.rsx_partition_count_device(self._h, d_src, n, ctypes.byref(lay), digit, nsub, d_hist, stream))

    def partition_scatter_device(self, d_src: int, d_dst: int, n: int, d: RadixDigits, digit: int, nsub: int, k: int,
                                 stream: int = 0):
        lay = d.layout()
        self._check(self._L.rsx_partition_scatter_device(self._h, d_src, d_dst, n, ctypes.byref(lay), digit, nsub, k, stream))

    def splitter_count_device(self, d_data: int, n: int, d: RadixDigits, d_ranges: int, d_prefix: int, nb: int, digit: int,
                              d_less: int, stream: int = 0):
        lay = d.layout()
        self._check(self._L.rsx_splitter_count_device(self._h, d_data, n, ctypes.byref(lay), d_ranges, d_prefix, nb, digit,
                                                      d_less, stream))

    def splitter_pick_device(self, d_total: int, d_rank: int, d_prefix: int, nb: int, digit: int, stream: int = 0):
        self._check(self._L.rsx_splitter_pick_device(self._h, d_total, d_rank, d_prefix, nb, digit, stream))

    def segmented_copy_device(self, d_src: int, d_dst: int, elem_bytes: int, d_src_off: int, d_dst_off: int,
                              d_len: int, nseg: int, stream: int = 0):
        self._check(self._L.rsx_segmented_copy_device(self._h, d_src, d_dst, elem_bytes, d_src_off, d_dst_off,
                                                      d_len, nseg, stream))

    def bounds_device(self, d_sorted: int, n: int, d: RadixDigits, d_queries: int, nq: int, d_out: int, stream: int = 0):
        lay = d.layout()
        self._check(self._L.rsx_bounds_device(self._h, d_sorted, n, ctypes.byref(lay), d_queries, nq, d_out, stream))

    def bounds_ranges_device(self, d_data: int, n: int, d: RadixDigits, d_queries: int, d_ranges: int, nq: int, d_out: int,
                             stream: int = 0):
        lay = d.layout()
        self._check(self._L.rsx_bounds_ranges_device(self._h, d_data, n, ctypes.byref(lay), d_queries, d_ranges, nq, d_out,
                                                     stream))

    def generate_device(self, d_data: int, n: int, d: RadixDigits, gen: int, seed: int, param: float = 0.0,
                        index_base: int = 0, stream: int = 0):
        lay = d.layout()
        self._check(self._L.rsx_generate_device(self._h, d_data, n, ctypes.byref(lay), gen, seed, param,
                                                index_base, stream))

    def verify_device(self, d_data: int, n: int, d: RadixDigits, d_out: int, stream: int = 0):
        lay = d.layout()
        self._check(self._L.rsx_verify_device(self._h, d_data, n, ctypes.byref(lay), d_out, stream))


_DEFAULT = {}
_DEFAULT_LOCK = threading.Lock()


def default_context(device: int) -> Context:
    with _DEFAULT_LOCK:
        c = _DEFAULT.get(device)
        if c is None:
            c = _DEFAULT[device] = Context(device)
        return c


def _torch_digits(t, digits: Optional[RadixDigits]) -> RadixDigits:
    if digits is not None:
        return digits
    import torch
    m = {torch.uint8: "u8", torch.int8: "i8", torch.int16: "i16", torch.int32: "i32", torch.int64: "i64",
         torch.float32: "f32", torch.float64: "f64"}
    for name, key in (("uint16", "u16"), ("uint32", "u32"), ("uint64", "u64")):
        if hasattr(torch, name):
            m[getattr(torch, name)] = key
    if t.dtype not in m:
        raise TypeError(f"no RadixDigits for torch dtype {t.dtype}; pass digits=")
    return PRIMITIVES[m[t.dtype]]


def radix_sort(x, digits: Optional[RadixDigits] = None, tmp=None, ctx: Optional[Context] = None):
    """`<[T]>::radix_sort(&mut self)` (mod.rs:62): sorts `x` in place and returns None.

    x: a contiguous torch tensor on a GPU (device-resident path, enqueued on the
       current stream, not synchronised), or a contiguous numpy array / CPU torch
       tensor (host drop-in path: H2D -> sort -> D2H, blocking).
    digits: RadixDigits of the element type; inferred for primitive dtypes and
       numpy structured dtypes (first field = key).  When given for a byte tensor
       (uint8), x is read as packed elements of digits.elem_bytes.
    tmp: optional ping-pong buffer of the same shape/dtype/device (mod.rs:71-83 `temp`).
    """
    if isinstance(x, np.ndarray):
        if not x.flags["C_CONTIGUOUS"] or not x.flags["WRITEABLE"]:
            raise ValueError("radix_sort needs a contiguous, writable array")
        d = digits if digits is not None else digits_of(x.dtype)
        n = x.nbytes // d.elem_bytes
        if x.nbytes % d.elem_bytes:
            raise ValueError("array size is not a multiple of elem_bytes")
        c = ctx or default_context(-1)
        c.sort_host(x.ctypes.data, n, d)
        return None
    import torch
    if not isinstance(x, torch.Tensor):
        raise TypeError("radix_sort expects a numpy array or a torch tensor")
    if not x.is_contiguous():
        raise ValueError("radix_sort needs a contiguous tensor")
    d = _torch_digits(x, digits)
    nbytes = x.numel() * x.element_size()
    if nbytes % d.elem_bytes:
        raise ValueError("tensor size is not a multiple of elem_bytes")
    n = nbytes // d.elem_bytes
    if not x.is_cuda:
        c = ctx or default_context(-1)
        c.sort_host(x.data_ptr(), n, d)
        return None
    dev = x.device.index if x.device.index is not None else torch.cuda.current_device()
    c = ctx or default_context(dev)
    if n <= 1:
        return None
    if tmp is None:
        tmp = torch.empty_like(x)
    elif tmp.device != x.device or tmp.numel() * tmp.element_size() < nbytes or not tmp.is_contiguous():
        raise ValueError("tmp must be a contiguous buffer on the same device, at least as large as x")
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        c.sort_device(x.data_ptr(), tmp.data_ptr(), n, d, stream)
    return None


def radix_sort_sharded(slices: Sequence, digits: RadixDigits, ctxs: Optional[Sequence[Context]] = None, tmps=None,
                       schedule: int = _lib.SHARD_EXCHANGE_FIRST):
    """rsx_sort_sharded: sorts the concatenation of `slices` (contiguous GPU tensors, one per
    context, possibly on different devices) as ONE array, stably and in place -- every slice
    keeps its length.  The single-process multi-GPU form of `<[T]>::radix_sort` with "chunk per
    thread" (mod.rs:66-70) read as "slice per GPU".  Blocking.  `schedule`: SHARD_EXCHANGE_FIRST (one
    partition pass by the top digit, exchange, one local sort) or SHARD_SORT_FIRST (sort, exchange, sort)."""
    import torch
    G = len(slices)
    if G == 0:
        return None
    if ctxs is None:
        ctxs = [Context(t.device.index if t.device.index is not None else torch.cuda.current_device()) for t in slices]
    if len(ctxs) != G:
        raise ValueError("one context per slice")
    if tmps is None:
        tmps = [torch.empty_like(t) for t in slices]
    ns = []
    for t, u in zip(slices, tmps):
        if not (t.is_cuda and t.is_contiguous() and u.is_cuda and u.is_contiguous() and u.device == t.device):
            raise ValueError("slices and tmps must be contiguous GPU tensors, pairwise on the same device")
        nbytes = t.numel() * t.element_size()
        if nbytes % digits.elem_bytes or u.numel() * u.element_size() < nbytes:
            raise ValueError("slice size is not a multiple of elem_bytes, or tmp too small")
        ns.append(nbytes // digits.elem_bytes)
    for t in slices:  # the library works on private streams: what torch enqueued must be done
        torch.cuda.synchronize(t.device)
    L = ctxs[0]._L
    hs = (ctypes.c_void_p * G)(*[c._h for c in ctxs])
    ps = (ctypes.c_void_p * G)(*[t.data_ptr() for t in slices])
    ts = (ctypes.c_void_p * G)(*[t.data_ptr() for t in tmps])
    nn = (ctypes.c_size_t * G)(*ns)
    lay = digits.layout()
    ctxs[0]._check(L.rsx_sort_sharded_ex(hs, G, ps, ts, nn, ctypes.byref(lay), schedule))
    return None
