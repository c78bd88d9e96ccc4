"""CPU: pins the C oracle (oracle/rsx_oracle.c) against the acceptance property of the
reference's own tests (src/radix_sort/tests.rs): integers == slice::sort (:7-23,25-131),
floats == sort_by(total_cmp) bitwise with +-0/NaN/+-inf present (:133-173), tuples ==
STABLE sort_by_key(.0) (:175-187) -- restated independently with numpy -- and against the
committed golden fixtures."""
import os
import struct

import numpy as np
import pytest

import util

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden.npz")


def _layout(orc, t):
    return orc.Layout(*util.TYPES[t])


@pytest.mark.parametrize("t", list(util.TYPES))
@pytest.mark.parametrize("dist", util.DISTS)
def test_oracle_matches_stable_sort(orc, t, dist):
    lay = _layout(orc, t)
    for n, threads in ((0, 1), (1, 8), (5, 8), (97, 3), (4099, 1), (20011, 3), (20011, 8)):
        raw = util.make_input(t, n, dist, seed=hash((t, dist, n)) & 0xFFFF)
        exp = orc.numpy_stable_sort(raw, lay)
        assert np.array_equal(orc.sort_parallel(raw, lay, threads), exp), (t, dist, n, threads)
    raw = util.make_input(t, 3001, dist, seed=7)
    assert np.array_equal(orc.sort0(raw, lay), orc.numpy_stable_sort(raw, lay))


@pytest.mark.parametrize("t", ["u8", "u32", "i64", "f32", "f64", "(u32,u32)"])
def test_reference_test_shapes_1e6(orc, t):
    """The reference's own test size: 1e6 elements (tests.rs:27,...)."""
    lay = _layout(orc, t)
    raw = util.make_input(t, 10 ** 6, "uniform", seed=11)
    got = orc.sort_parallel(raw, lay, threads=8)
    assert np.array_equal(got, orc.numpy_stable_sort(raw, lay))


def test_float_total_order(orc):
    """-NaN < -inf < -1 < -0.0 < +0.0 < 1 < +inf < +NaN (f32::total_cmp; tests.rs:146-151)."""
    vals = [float("nan"), float("inf"), 1.0, 0.0, -0.0, -1.0, float("-inf")]
    bits = [struct.unpack("<I", struct.pack("<f", v))[0] for v in vals] + [0xFFC00000]  # -NaN
    raw = np.array(bits, dtype="<u4").view(np.uint8)
    got = orc.sort0(raw, orc.Layout(4, 0, 4, util.FLOAT)).view("<u4").tolist()
    assert got == [0xFFC00000, 0xFF800000, 0xBF800000, 0x80000000, 0x00000000, 0x3F800000, 0x7F800000, 0x7FC00000]


def test_signed_order(orc):
    raw = np.array([0, -1, 127, -128, 5], dtype=np.int8).view(np.uint8)
    assert orc.sort0(raw, orc.Layout(1, 0, 1, util.SIGNED)).view(np.int8).tolist() == [-128, -1, 0, 5, 127]


def test_get_digit_examples(orc):
    """Spot values of radix_digits.rs get_digit."""
    import ctypes
    L = orc.lib()

    def dig(b, lay, i):
        buf = (ctypes.c_uint8 * len(b))(*b)
        return L.orc_get_digit(ctypes.addressof(buf), ctypes.byref(lay), i)

    u32 = orc.Layout(4, 0, 4, 0)
    assert [dig(list(struct.pack("<I", 0x11223344)), u32, i) for i in range(4)] == [0x44, 0x33, 0x22, 0x11]
    i32 = orc.Layout(4, 0, 4, 1)
    assert [dig(list(struct.pack("<i", -1)), i32, i) for i in range(4)] == [0xFF, 0xFF, 0xFF, 0x7F]
    f32 = orc.Layout(4, 0, 4, 2)
    assert [dig(list(struct.pack("<f", 1.0)), f32, i) for i in range(4)] == [0x00, 0x00, 0x80, 0xBF]
    assert [dig(list(struct.pack("<f", -1.0)), f32, i) for i in range(4)] == [0xFF, 0xFF, 0x7F, 0x40]
    tup = orc.Layout(8, 4, 4, 0)  # key after a 4-byte payload
    assert dig([9, 9, 9, 9, 1, 2, 3, 4], tup, 2) == 3


def test_partition_pass_is_stable_by_digit(orc):
    lay = _layout(orc, "(u32,u32)")
    raw = util.make_input("(u32,u32)", 5000, "uniform", seed=3)
    out, hist = orc.partition_pass(raw, lay, 1)
    e = raw.reshape(-1, 8)
    order = np.argsort(e[:, 1], kind="stable")
    assert np.array_equal(out.reshape(-1, 8), e[order])
    assert np.array_equal(hist, np.bincount(e[:, 1], minlength=256).astype(np.uint64))


def test_golden_fixtures(orc):
    """Every committed (input, expected) pair: oracle (both forms) and numpy agree with it."""
    z = np.load(GOLDEN)
    keys = [k[3:] for k in z.files if k.startswith("in|")]
    assert len(keys) > 2000
    for k in keys:
        t, dist, n, seed = k.split("|")
        lay = _layout(orc, t)
        raw, exp = z["in|" + k], z["out|" + k]
        assert raw.size == int(n) * lay.elem_bytes
        assert np.array_equal(orc.sort_parallel(raw, lay, 5), exp), k
        if int(n) <= 1000:
            assert np.array_equal(orc.numpy_stable_sort(raw, lay), exp), k
        assert np.array_equal(util.make_input(t, int(n), dist, int(seed)), raw), "fixture input not reproducible: " + k


# ---- the reference's optimisation ladder radix_sort0..5 + counting_sort (mod.rs:40-59,178-571) ----
@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("t", ["u8", "u16", "u32", "i64", "f32", "(u64,u64)", "(u8,u8)", "u128", "(u32,[u8;8])"])
def test_cpu_ladder_variants_agree(orc, t, variant):
    lay = orc.Layout(*util.TYPES[t])
    for n, dist, threads in ((1, "uniform", 3), (2, "uniform", 8), (1000, "two", 7), (65537, "uniform", 5), (40001, "zipf", 2)):
        raw = util.make_input(t, n, dist, seed=n + variant)
        assert np.array_equal(orc.sort_variant(raw, lay, threads, variant), orc.numpy_stable_sort(raw, lay)), (t, variant, n, dist)


def test_counting_sort_bytes(orc):
    rng = np.random.default_rng(3)
    for n in (0, 1, 255, 100003):
        a = rng.integers(0, 256, size=n, dtype=np.uint8)
        assert np.array_equal(orc.counting_sort(a), np.sort(a, kind="stable"))
