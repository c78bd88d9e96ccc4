"""CPU: host-side logic of the Python mirror (descriptors, dtype mapping)."""
import numpy as np
import pytest

import radix_sort_amd as rs
import util


def test_primitive_number_of_digits():
    # radix_digits.rs: NUMBER_OF_DIGITS per type
    exp = {"u8": 1, "u16": 2, "u32": 4, "u64": 8, "u128": 16, "usize": 8, "i8": 1, "i16": 2, "i32": 4, "i64": 8,
           "i128": 16, "isize": 8, "f32": 4, "f64": 8}
    for k, d in exp.items():
        assert rs.PRIMITIVES[k].NUMBER_OF_DIGITS == d
        assert rs.PRIMITIVES[k].elem_bytes == (8 if k.endswith("size") else int(k[1:]) // 8)


def test_digits_of_numpy_dtypes():
    assert rs.digits_of(np.uint32) == rs.PRIMITIVES["u32"]
    assert rs.digits_of(np.int64) == rs.PRIMITIVES["i64"]
    assert rs.digits_of(np.float32) == rs.PRIMITIVES["f32"]
    pair = np.dtype([("k", "<u8"), ("v", "<u8")])
    assert rs.digits_of(pair) == rs.RadixDigits(16, 0, 8, rs.KEY_UNSIGNED)
    swapped = np.dtype({"names": ["k", "v"], "formats": ["<f8", "<u8"], "offsets": [8, 0], "itemsize": 16})
    assert rs.digits_of(swapped) == rs.RadixDigits(16, 8, 8, rs.KEY_FLOAT)
    with pytest.raises(TypeError):
        rs.digits_of(np.complex64)


def test_tuple_of_default_layout():
    assert rs.tuple_of("u32", 4) == rs.RadixDigits(8, 0, 4, rs.KEY_UNSIGNED)
    assert rs.tuple_of("u64", 8) == rs.RadixDigits(16, 0, 8, rs.KEY_UNSIGNED)
    assert rs.tuple_of("u32", 1) == rs.RadixDigits(8, 0, 4, rs.KEY_UNSIGNED)  # padded to key alignment
    assert rs.tuple_of("f64", 8, key_offset=8, elem_bytes=16) == rs.RadixDigits(16, 8, 8, rs.KEY_FLOAT)


@pytest.mark.parametrize("t", list(util.TYPES))
def test_get_digit_matches_oracle(orc, t):
    import ctypes
    es, ko, kb, kind = util.TYPES[t]
    d = rs.RadixDigits(es, ko, kb, kind)
    lay = orc.Layout(es, ko, kb, kind)
    raw = util.make_input(t, 64, "uniform", 5)
    L = orc.lib()
    for i in range(64):
        e = raw[i * es:(i + 1) * es]
        for idx in range(kb):
            assert d.get_digit(bytes(e), idx) == L.orc_get_digit(e.ctypes.data, ctypes.byref(lay), idx)


def test_radix_sort_rejects_bad_input():
    with pytest.raises(TypeError):
        rs.radix_sort([3, 1, 2])
    with pytest.raises(ValueError):
        rs.radix_sort(np.arange(10, dtype=np.uint32)[::2])
