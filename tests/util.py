"""Shared input builders for the parity tests (seeded, numpy only)."""
from __future__ import annotations

import numpy as np

UNSIGNED, SIGNED, FLOAT = 0, 1, 2

# name -> (elem_bytes, key_offset, key_bytes, key_kind): the 14 built-in key kinds of
# radix_digits.rs:7-124 (usize/isize = 64-bit) plus (T,U) tuples (:126-136)
TYPES = {
    "u8": (1, 0, 1, UNSIGNED), "u16": (2, 0, 2, UNSIGNED), "u32": (4, 0, 4, UNSIGNED),
    "u64": (8, 0, 8, UNSIGNED), "u128": (16, 0, 16, UNSIGNED), "usize": (8, 0, 8, UNSIGNED),
    "i8": (1, 0, 1, SIGNED), "i16": (2, 0, 2, SIGNED), "i32": (4, 0, 4, SIGNED),
    "i64": (8, 0, 8, SIGNED), "i128": (16, 0, 16, SIGNED), "isize": (8, 0, 8, SIGNED),
    "f32": (4, 0, 4, FLOAT), "f64": (8, 0, 8, FLOAT),
    # tuples: (key, payload) -- key first
    "(u32,u32)": (8, 0, 4, UNSIGNED), "(u64,u64)": (16, 0, 8, UNSIGNED),
    "(u8,u8)": (2, 0, 1, UNSIGNED), "(u8,[u8;7])": (8, 0, 1, UNSIGNED), "(i16,u16)": (4, 0, 2, SIGNED),
    "(f32,u32)": (8, 0, 4, FLOAT), "(u32,[u8;8])": (12, 0, 4, UNSIGNED),
    "(u64,[u64;2])": (24, 0, 8, UNSIGNED), "(u128,u128)": (32, 0, 16, UNSIGNED),
    # key NOT first (rustc may reorder tuple fields): payload u32 then key u32; payload u64 then f64 key
    "(pay32+u32)": (8, 4, 4, UNSIGNED), "(pay64+f64)": (16, 8, 8, FLOAT),
}
PRIMS = [k for k in TYPES if not k.startswith("(")]
TUPLES = [k for k in TYPES if k.startswith("(")]

DISTS = ["uniform", "equal", "sorted", "reversed", "two", "lowbyte", "highbyte", "zipf", "step16"]


def _key_ints(dist: str, n: int, key_bytes: int, rng) -> np.ndarray:
    """(n, key_bytes) uint8 little-endian raw key bit patterns."""
    bits = key_bytes * 8
    if dist == "uniform":
        return rng.integers(0, 256, size=(n, key_bytes), dtype=np.uint8)
    out = np.zeros((n, key_bytes), dtype=np.uint8)

    def put(vals):  # vals: python ints or uint64 array (< 2^64) -> low 8 bytes
        v = np.asarray(vals, dtype=np.uint64)
        b = v.view(np.uint8).reshape(n, 8)
        out[:, : min(8, key_bytes)] = b[:, : min(8, key_bytes)]

    if dist == "equal":
        out[:] = rng.integers(0, 256, size=(1, key_bytes), dtype=np.uint8)
    elif dist == "sorted":
        put(np.arange(n, dtype=np.uint64) & np.uint64((1 << min(bits, 63)) - 1))
    elif dist == "reversed":
        put((np.uint64(n) - 1 - np.arange(n, dtype=np.uint64)) & np.uint64((1 << min(bits, 63)) - 1))
    elif dist == "two":
        pats = rng.integers(0, 256, size=(2, key_bytes), dtype=np.uint8)
        out[:] = pats[rng.integers(0, 2, size=n)]
    elif dist == "lowbyte":
        out[:, 0] = rng.integers(0, 256, size=n, dtype=np.uint8)
    elif dist == "highbyte":
        out[:, key_bytes - 1] = rng.integers(0, 256, size=n, dtype=np.uint8)
    elif dist == "zipf":  # log-uniform over [1, 2^min(bits,63)): Zipf(s=1)-shaped (distr.rs:54-76)
        u = rng.random(n)
        put(np.floor(np.exp(u * np.log(2.0 ** min(bits, 63)))).astype(np.uint64))
    elif dist == "step16":  # 16 equally spaced values (distr.rs:78-106)
        maxv = (1 << min(bits, 64)) - 1
        s = maxv // 17
        vals = np.array([s * (i + 1) for i in range(16)], dtype=np.uint64)
        put(vals[rng.integers(0, 16, size=n)])
    else:
        raise ValueError(dist)
    return out


def make_input(tname: str, n: int, dist: str, seed: int) -> np.ndarray:
    """Raw little-endian bytes (uint8, n*elem_bytes) of a seeded input."""
    es, ko, kb, kind = TYPES[tname]
    rng = np.random.default_rng(seed)
    raw = np.zeros((n, es), dtype=np.uint8)
    if n == 0:
        return raw.reshape(-1)
    raw[:, ko:ko + kb] = _key_ints(dist, n, kb, rng)
    if kind == FLOAT and dist == "uniform" and n >= 16:
        # what tests.rs:135-143 injects: 0.0, -0.0, NaN, +inf, -inf (+ extra NaN payloads, denormals)
        if kb == 4:
            specials = np.array([0x00000000, 0x80000000, 0x7FC00000, 0x7F800000, 0xFF800000, 0xFFC00001,
                                 0x7F800001, 0x00000001, 0x80000001, 0x7FFFFFFF, 0xFFFFFFFF], dtype="<u4")
        else:
            specials = np.array([0x0, 0x8000000000000000, 0x7FF8000000000000, 0x7FF0000000000000,
                                 0xFFF0000000000000, 0xFFF8000000000001, 0x7FF0000000000001, 0x1,
                                 0x8000000000000001, 0x7FFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF], dtype="<u8")
        pos = rng.choice(n, size=min(len(specials), n), replace=False)
        sb = specials.view(np.uint8).reshape(len(specials), kb)
        raw[pos, ko:ko + kb] = sb[: len(pos)]
    # payload: original index (reveals instability), little-endian, truncated/zero-extended
    pay = [b for b in range(es) if not (ko <= b < ko + kb)]
    if pay:
        idx = np.arange(n, dtype=np.uint64).view(np.uint8).reshape(n, 8)
        for j, b in enumerate(pay):
            raw[:, b] = idx[:, j] if j < 8 else 0
    return raw.reshape(-1)


def layout_tuple(tname: str):
    return TYPES[tname]
