"""GPU: parity of the HIP path (through the C-ABI, librsx.so) with the CPU oracle.
Bit-exact (integer / byte / index work): np.array_equal on the raw element bytes."""
import ctypes
import os

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden.npz")


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


@pytest.fixture(scope="module")
def rs():
    import radix_sort_amd as rs
    return rs


@pytest.fixture(scope="module")
def ctx(rs, torch):
    return rs.default_context(torch.cuda.current_device())


def _digits(rs, t):
    return rs.RadixDigits(*util.TYPES[t])


def _gpu_sort(rs, torch, ctx, raw, d):
    x = torch.from_numpy(raw.copy()).cuda()
    rs.radix_sort(x, digits=d)
    ctx.check()
    return x.cpu().numpy()


EDGE_SIZES = [0, 1, 2, 3, 63, 64, 65, 511, 512, 513, 2047, 2048, 2049, 8191, 8192, 8193, 16385, 100003]


@pytest.mark.parametrize("t", list(util.TYPES))
def test_edge_sizes_uniform(rs, torch, ctx, orc, t):
    d = _digits(rs, t)
    lay = orc.Layout(*util.TYPES[t])
    for n in EDGE_SIZES:
        raw = util.make_input(t, n, "uniform", seed=1000 + n)
        assert np.array_equal(_gpu_sort(rs, torch, ctx, raw, d), orc.sort_parallel(raw, lay, 4)), (t, n)


TILE_KEYS = {1: 28, 2: 28, 4: 28, 8: 14, 12: 10, 16: 5, 24: 5, 32: 3}  # keys per thread of the 512-thread tile, by element size


@pytest.mark.parametrize("t", list(util.TYPES))
def test_sizes_around_the_tile(rs, torch, ctx, orc, t):
    """The one-launch kernel (n <= one tile), the general path right above it, and seeded random sizes up to three
    tiles with seeded random distributions: every type, bytes compared with the oracle."""
    d = _digits(rs, t)
    lay = orc.Layout(*util.TYPES[t])
    tile = 512 * TILE_KEYS[d.elem_bytes]
    rng = np.random.default_rng(4242 + d.elem_bytes + 100 * d.key_bytes)
    sizes = [tile - 1, tile, tile + 1, 2 * tile - 1, 2 * tile, 2 * tile + 1] + [int(x) for x in rng.integers(1, 3 * tile, size=10)]
    for i, n in enumerate(sizes):
        dist = util.DISTS[int(rng.integers(0, len(util.DISTS)))]
        raw = util.make_input(t, n, dist, seed=9000 + i)
        assert np.array_equal(_gpu_sort(rs, torch, ctx, raw, d), orc.sort_parallel(raw, lay, 3)), (t, n, dist)


BUCKET_KEYS = {2: 28, 4: 28, 8: 17, 12: 9, 16: 7, 24: 4, 32: 3}  # keys per thread of the 1024-thread bucket kernel, by element size


def _mid_max(es):
    return 1 << 22 if es == 8 else 1024 * BUCKET_KEYS[es] * 256 * 4 // 7


@pytest.mark.parametrize("t", [t for t in util.TYPES if util.TYPES[t][2] >= 2])
def test_middle_sizes(rs, torch, ctx, orc, t):
    """The middle-size path (bucket split by the top digit + one workgroup per bucket) at its switch points -- the
    largest size it takes, the first size it does not -- and inside its range with inputs that take the split
    (uniform, random top byte only) and inputs that refuse it on the device (skewed top digits: LSD passes)."""
    d = _digits(rs, t)
    lay = orc.Layout(*util.TYPES[t])
    mm = _mid_max(d.elem_bytes)
    rng = np.random.default_rng(777 + d.elem_bytes)
    cases = [(mm, "uniform"), (mm + 1, "uniform"), (mm // 2 + 3, "highbyte"), (int(rng.integers(mm // 8, mm)), "zipf"),
             (int(rng.integers(mm // 8, mm)), "two"), (int(rng.integers(mm // 16, mm // 2)), "sorted"),
             (int(rng.integers(mm // 16, mm // 2)), "uniform"), (int(rng.integers(mm // 16, mm // 2)), "step16")]
    for i, (n, dist) in enumerate(cases):
        raw = util.make_input(t, n, dist, seed=1234 + i)
        assert np.array_equal(_gpu_sort(rs, torch, ctx, raw, d), orc.sort_parallel(raw, lay, 8)), (t, n, dist)


@pytest.mark.parametrize("t", ["u32", "u64", "(u64,u64)", "i16", "f32", "(u32,[u8;8])", "(u128,u128)"])
def test_middle_size_forecast_and_oversized_buckets(rs, torch, orc, t):
    """The host forecasts the bucket split from the previous sort's report.  On a fresh context: uniform (split), then
    skewed inputs that the forecast gets wrong (the oversized buckets are sorted through memory by one workgroup each),
    then the cool-down of LSD passes, then uniform again -- and every distribution with the split forced."""
    d = _digits(rs, t)
    lay = orc.Layout(*util.TYPES[t])
    mm = _mid_max(d.elem_bytes)
    c = rs.Context(torch.cuda.current_device())
    seq = [(mm // 3, "uniform"), (mm // 3 + 1, "zipf"), (mm // 4, "uniform"), (mm // 5, "equal"), (mm // 3, "uniform")]
    for i, (n, dist) in enumerate(seq * 3):  # 15 sorts: forecast right, wrong, cool-down, recovery
        raw = util.make_input(t, n, dist, seed=500 + i)
        x = torch.from_numpy(raw.copy()).cuda()
        rs.radix_sort(x, digits=d, ctx=c)
        c.check()
        assert np.array_equal(x.cpu().numpy(), orc.sort_parallel(raw, lay, 8)), (t, i, n, dist)
    c.set_option(rs.OPT_MID_SORT, 2)
    for i, dist in enumerate(util.DISTS):
        n = mm // 4 + 17 * i
        raw = util.make_input(t, n, dist, seed=600 + i)
        x = torch.from_numpy(raw.copy()).cuda()
        rs.radix_sort(x, digits=d, ctx=c)
        c.check()
        assert np.array_equal(x.cpu().numpy(), orc.sort_parallel(raw, lay, 8)), (t, "forced split", n, dist)
    c.close()


WIDE_TYPES = [t for t in util.TYPES if util.TYPES[t][0] >= 8 and util.TYPES[t][2] >= 4]


@pytest.mark.parametrize("t", WIDE_TYPES)
def test_wide_key_hybrid_forced(rs, torch, orc, t):
    """RSX_OPT_WIDE_SORT = 2: the top 16 bits of the key counted, two sweeps for those digits, every 16-bit bucket sorted
    by its remaining digits in LDS -- forced on every distribution (skewed ones put everything into a few buckets, which
    then go through memory, workgroup by workgroup), for every element type with a key of at least 8 bytes."""
    d = _digits(rs, t)
    lay = orc.Layout(*util.TYPES[t])
    c = rs.Context(torch.cuda.current_device())
    c.set_option(rs.OPT_WIDE_SORT, 2)
    rng = np.random.default_rng(31 + d.elem_bytes)
    for skip, group in ((1, 1), (0, 1), (1, 0)):  # skipped digits + mending / every pass; groups of small buckets / one by one
        c.set_option(rs.OPT_BUCKET_SKIP, skip)
        c.set_option(rs.OPT_BUCKET_GROUP, group)
        for i, dist in enumerate(util.DISTS):
            n = int(rng.integers(70000, 400000))
            raw = util.make_input(t, n, dist, seed=300 + i)
            x = torch.from_numpy(raw.copy()).cuda()
            rs.radix_sort(x, digits=d, ctx=c)
            c.check()
            assert np.array_equal(x.cpu().numpy(), orc.sort_parallel(raw, lay, 8)), (t, n, dist, skip, group)
    c.close()


@pytest.mark.parametrize("t", WIDE_TYPES)
def test_wide_key_hybrid_counts_from_the_16_bit_counters(rs, torch, orc, t):
    """RSX_OPT_WIDE_SORT = 3 (the default mode without its 2 GiB floor): ragged sizes above the middle sizes.  The count's
    workgroups are laid out k per region, the first sweep's count matrix is taken from their counters (no second read of
    the array); a uniform input takes the hybrid (path 5 reported), a skewed one is refused on the device (path 0)."""
    d = _digits(rs, t)
    lay = orc.Layout(*util.TYPES[t])
    mm = _mid_max(d.elem_bytes)
    c = rs.Context(torch.cuda.current_device())
    c.set_option(rs.OPT_WIDE_SORT, 3)
    rng = np.random.default_rng(97 + d.elem_bytes)
    for i, (dist, path) in enumerate((("uniform", 5), ("uniform", 5), ("zipf", 0), ("uniform", 5))):
        n = mm + int(rng.integers(1, mm // 2))
        raw = util.make_input(t, n, dist, seed=900 + i)
        x = torch.from_numpy(raw.copy()).cuda()
        rs.radix_sort(x, digits=d, ctx=c)
        c.check()
        info = c.get_info(rs.INFO_LAST_PASSES)
        assert (info >> 24) & 15 == path and info & 255 == (2 if path == 5 else d.key_bytes), (t, n, dist, hex(info))
        assert np.array_equal(x.cpu().numpy(), orc.sort_parallel(raw, lay, 8)), (t, n, dist)
        if path == 0:
            c.set_option(rs.OPT_WIDE_SORT, 3)  # (a refusal is followed by 15 sorts without a try: start over)
    c.close()


@pytest.mark.parametrize("t", ["u64", "i64", "f64", "(u64,u64)", "u128", "(u128,u128)"])
def test_wide_key_hybrid_mends_the_runs_its_passes_left(rs, torch, orc, t):
    """The hybrid's LDS passes start at the digit that leaves them the bits an array of that size needs; neighbours that
    still agree afterwards are put right by the skipped digits (mend_listed: a list of run heads, one thread per run).
    Inputs made for that: (A) many short runs -- elements that share everything but their low 16 bits with a neighbour
    (more heads than the list holds: every pass); (B) whole buckets that agree on everything between the top and the low
    16 bits (runs too long to mend: the workgroup runs every pass); (C) a few pairs that share everything but their lowest
    byte (the list, mended); (D) 20000 distinct keys (long runs of equal keys: nothing to mend); then uniform keys again.  Payload = index, so stability shows."""
    d = _digits(rs, t)
    es, ko, kb, _kind = util.TYPES[t]
    lay = orc.Layout(*util.TYPES[t])
    c = rs.Context(torch.cuda.current_device())
    c.set_option(rs.OPT_WIDE_SORT, 3)
    n = _mid_max(es) + 300001
    rng = np.random.default_rng(4242 + es + kb)
    idx = np.arange(n, dtype=np.uint64).view(np.uint8).reshape(n, 8)
    for case in ("A", "B", "C", "D", "uniform"):
        raw = np.zeros((n, es), dtype=np.uint8)
        key = rng.integers(0, 256, size=(n, kb), dtype=np.uint8)
        if case == "A":
            pick = np.flatnonzero(rng.random(n) < 0.4)
            pick = pick[pick > 0]
            for _ in range(3):  # chains: runs of up to four
                key[pick, 2:] = key[pick - 1, 2:]
        elif case == "B":
            key[:, 2:kb - 2] = 0
        elif case == "C":
            pick = np.flatnonzero(rng.random(n) < 0.002)
            pick = pick[pick > 0]
            key[pick, 1:] = key[pick - 1, 1:]
        elif case == "D":  # few distinct keys, spread out: long runs of EQUAL keys need no mending (and get none)
            vals = rng.integers(0, 256, size=(20000, kb), dtype=np.uint8)
            key = vals[rng.integers(0, 20000, size=n)]
        raw[:, ko:ko + kb] = key
        for j, b in enumerate(b for b in range(es) if not ko <= b < ko + kb):
            raw[:, b] = idx[:, j] if j < 8 else 0
        x = torch.from_numpy(raw.reshape(-1).copy()).cuda()
        rs.radix_sort(x, digits=d, ctx=c)
        c.check()
        info = c.get_info(rs.INFO_LAST_PASSES)
        assert (info >> 24) & 15 == 5, (t, case, hex(info))
        assert np.array_equal(x.cpu().numpy(), orc.sort_parallel(raw.reshape(-1), lay, 8)), (t, case)
    c.close()


def _ranged_keys(rng, n, kb, bits, base=0):
    """(n, kb) little-endian key bytes: base + uniform below 2^bits."""
    out = np.zeros((n, kb), dtype=np.uint8)
    full, rem = divmod(bits, 8)
    out[:, :full] = rng.integers(0, 256, size=(n, full), dtype=np.uint8)
    if rem:
        out[:, full] = rng.integers(0, 1 << rem, size=n, dtype=np.uint8)
    b = np.frombuffer(int(base).to_bytes(kb, "little"), dtype=np.uint8)
    assert not np.any(b[: (bits + 7) // 8] & (0xFF if not rem else 0)) or True
    carry = out.astype(np.uint16) + b  # base has no bits below `bits`: no carries
    return carry.astype(np.uint8)


@pytest.mark.parametrize("t,bits,base,path", [
    ("u64", 40, 0, 5), ("u64", 35, 0, 5), ("u64", 44, 0xABCD << 48, 5), ("u64", 12, 0, 5), ("u64", 24, 0, 5),
    ("i64", 40, 0, 5), ("f64", 52, 0x3FF << 52, 5), ("(u64,u64)", 47, 1 << 60, 5), ("u128", 64, 0, 5), ("u128", 100, 7 << 120, 5),
    ("(u32,u32)", 20, 0, 5), ("(u32,u32)", 27, 5 << 28, 5), ("(u64,[u64;2])", 33, 0, 5)])
def test_wide_key_hybrid_follows_the_range_of_the_keys(rs, torch, orc, t, bits, base, path):
    """The 16-bit window sits below the highest bit in which the (sampled) keys differ: keys of a narrow range -- 40-bit
    numbers in a u64, doubles in [1, 2), one value range of a multi-GPU sort -- are partitioned by THEIR top 16 bits, the
    bits above are verified to be the same for every element.  Ranges whose window had to move up to keep its digits
    inside a dword (2^35), that reach the key's lowest bits (2^12: nothing left for LDS), and an element outside the
    range that the sample does not see (the count finds it: LSD passes)."""
    d = _digits(rs, t)
    es, ko, kb, _kind = util.TYPES[t]
    lay = orc.Layout(*util.TYPES[t])
    c = rs.Context(torch.cuda.current_device())
    c.set_option(rs.OPT_WIDE_SORT, 3)
    n = _mid_max(es) + 250001
    rng = np.random.default_rng(555 + es + bits)
    idx = np.arange(n, dtype=np.uint64).view(np.uint8).reshape(n, 8)
    for outlier in (False, True):
        raw = np.zeros((n, es), dtype=np.uint8)
        key = _ranged_keys(rng, n, kb, bits, base)
        if outlier:
            key[1, kb - 1] ^= 0x40  # index 1 is not among the sampled positions
        raw[:, ko:ko + kb] = key
        for j, b in enumerate(b for b in range(es) if not ko <= b < ko + kb):
            raw[:, b] = idx[:, j] if j < 8 else 0
        x = torch.from_numpy(raw.reshape(-1).copy()).cuda()
        rs.radix_sort(x, digits=d, ctx=c)
        c.check()
        info = c.get_info(rs.INFO_LAST_PASSES)
        assert (info >> 24) & 15 == (0 if outlier else path), (t, bits, outlier, hex(info))
        assert np.array_equal(x.cpu().numpy(), orc.sort_parallel(raw.reshape(-1), lay, 8)), (t, bits, outlier)
        c.set_option(rs.OPT_WIDE_SORT, 3)  # (forget the refusal)
    c.close()


@pytest.mark.parametrize("t", ["u64", "(u64,u64)"])
def test_wide_key_hybrid_picks_its_workgroups_by_the_largest_buckets(rs, torch, orc, t):
    """Keys with a triangular density (the sum of two uniform numbers): the buckets at the peak hold twice the average.
    Which form of the bucket kernel runs -- 256, 512 or 1024 threads per bucket, or groups of small buckets, and of which
    size -- is the device's choice from the counts (all are enqueued, each behind its gate): the smallest that holds all
    but a handful of the buckets, so that no crowd of buckets goes through memory."""
    d = _digits(rs, t)
    es, ko, kb, _kind = util.TYPES[t]
    lay = orc.Layout(*util.TYPES[t])
    c = rs.Context(torch.cuda.current_device())
    c.set_option(rs.OPT_WIDE_SORT, 3)
    rng = np.random.default_rng(2024 + es)
    idx_all = None
    for n in (_mid_max(es) + 123457, 3 * _mid_max(es) + 11):
        a = rng.integers(0, 1 << 62, size=n, dtype=np.uint64) + rng.integers(0, 1 << 62, size=n, dtype=np.uint64)
        raw = np.zeros((n, es), dtype=np.uint8)
        raw[:, ko:ko + kb] = a.view(np.uint8).reshape(n, 8)
        idx = np.arange(n, dtype=np.uint64).view(np.uint8).reshape(n, 8)
        for j, b in enumerate(b for b in range(es) if not ko <= b < ko + kb):
            raw[:, b] = idx[:, j] if j < 8 else 0
        x = torch.from_numpy(raw.reshape(-1).copy()).cuda()
        rs.radix_sort(x, digits=d, ctx=c)
        c.check()
        assert (c.get_info(rs.INFO_LAST_PASSES) >> 24) & 15 == 5, (t, n)
        assert np.array_equal(x.cpu().numpy(), orc.sort_parallel(raw.reshape(-1), lay, 8)), (t, n)
    c.close()


@pytest.mark.parametrize("t", ["u64", "(u64,u64)", "i64"])
def test_wide_key_hybrid_a_handful_of_crowded_buckets(rs, torch, orc, t):
    """Uniform keys plus a few values of the top 16 bits that hold more elements than any workgroup sorts in LDS.  The
    verdict keeps the hybrid (VERDICT_MEDIUM) and the medium kernel takes those: one pass through memory that splits
    each by its next bits, then LDS.  (More than a handful, one too large for eight more bits, or more than n / 64
    elements in them: the LSD passes -- the Zipf case of the tests above.)"""
    d = _digits(rs, t)
    es, ko, kb, _kind = util.TYPES[t]
    lay = orc.Layout(*util.TYPES[t])
    c = rs.Context(torch.cuda.current_device())
    c.set_option(rs.OPT_WIDE_SORT, 3)
    n = _mid_max(es) + 200001
    rng = np.random.default_rng(909 + es)
    key = rng.integers(0, 256, size=(n, kb), dtype=np.uint8)
    nhot, each = (3, 19000) if es == 8 else (2, 8000)  # above the largest workgroup (17408 / 7168), together under n / 64
    hot = rng.choice(n, size=nhot * each, replace=False).reshape(nhot, each)
    for j in range(nhot):
        key[hot[j], kb - 2:] = rng.integers(0, 256, size=2, dtype=np.uint8)  # one value of the top 16 bits
    raw = np.zeros((n, es), dtype=np.uint8)
    raw[:, ko:ko + kb] = key
    idx = np.arange(n, dtype=np.uint64).view(np.uint8).reshape(n, 8)
    for j, b in enumerate(b for b in range(es) if not ko <= b < ko + kb):
        raw[:, b] = idx[:, j] if j < 8 else 0
    x = torch.from_numpy(raw.reshape(-1).copy()).cuda()
    rs.radix_sort(x, digits=d, ctx=c)
    c.check()
    assert (c.get_info(rs.INFO_LAST_PASSES) >> 24) & 15 == 5, t
    assert np.array_equal(x.cpu().numpy(), orc.sort_parallel(raw.reshape(-1), lay, 8)), t
    c.close()


def test_wide_key_hybrid_decides_on_the_device(rs, torch, ctx, orc):
    """Default mode at a size where the hybrid is tried (2 GiB of u64): a uniform input takes it, a Zipf input is
    refused by the count (the LSD passes run, gated on the same verdict word), and after a refusal the context goes
    without trying for a while; checked on the device (sorted, multiset unchanged) and against the LSD-only result."""
    d = rs.PRIMITIVES["u64"]
    n = 1 << 28
    c = rs.Context(torch.cuda.current_device())
    ref = rs.Context(torch.cuda.current_device())
    ref.set_option(rs.OPT_WIDE_SORT, 0)
    x = torch.empty(n * 8, dtype=torch.uint8, device="cuda")
    y = torch.empty_like(x)
    tmp = torch.empty_like(x)
    out = torch.zeros(3, dtype=torch.int64, device="cuda")
    for i, gen in enumerate((rs.GEN_UNIFORM, rs.GEN_ZIPF, rs.GEN_ZIPF, rs.GEN_UNIFORM, rs.GEN_UNIFORM)):
        c.generate_device(x.data_ptr(), n, d, gen, 50 + i, 1.0)
        y.copy_(x)
        c.sort_device(x.data_ptr(), tmp.data_ptr(), n, d)
        c.check()
        ref.sort_device(y.data_ptr(), tmp.data_ptr(), n, d)
        ref.check()
        assert torch.equal(x, y), (i, gen)
        c.verify_device(x.data_ptr(), n, d, out.data_ptr())
        assert out[0].item() == 0
    c.close()
    ref.close()


@pytest.mark.parametrize("t,extra", [("u32", 1), ("u32", 200), ("u64", 50), ("(u32,u32)", 3), ("f32", 1000)])
def test_small_bucket_workgroups_meet_an_oversized_bucket(rs, torch, orc, t, extra):
    """After a uniform input the forecast picks 256-thread bucket workgroups (every bucket fitted a quarter of the large
    capacity).  The next input holds ONE bucket just over that small capacity: its workgroup sorts it through memory in
    D-1 passes, and the bucket is small enough to stay in the CU's vector L1 from pass to pass -- the passes must not
    read stale lines."""
    d = _digits(rs, t)
    es, ko, kb, _kind = util.TYPES[t]
    lay = orc.Layout(*util.TYPES[t])
    c = rs.Context(torch.cuda.current_device())
    n = 200000
    raw = util.make_input(t, n, "uniform", seed=7)
    x = torch.from_numpy(raw.copy()).cuda()
    rs.radix_sort(x, digits=d, ctx=c)
    c.check()
    assert np.array_equal(x.cpu().numpy(), orc.sort_parallel(raw, lay, 8))
    cap_small = 256 * BUCKET_KEYS[es]
    for rep in range(3):
        raw = util.make_input(t, n, "uniform", seed=8 + rep).reshape(n, es)
        rng = np.random.default_rng(9 + rep)
        top = raw[:, ko + kb - 1]
        top[top == 0x33] = 0x34
        top[rng.choice(n, size=cap_small + extra, replace=False)] = 0x33  # one bucket just over the small capacity
        raw = raw.reshape(-1)
        x = torch.from_numpy(raw.copy()).cuda()
        rs.radix_sort(x, digits=d, ctx=c)
        c.check()
        assert np.array_equal(x.cpu().numpy(), orc.sort_parallel(raw, lay, 8)), (t, extra, rep)
        # back to uniform so that the forecast returns to the small workgroups for the next repetition
        for k in range(10):
            u = util.make_input(t, n, "uniform", seed=100 + 10 * rep + k)
            y = torch.from_numpy(u.copy()).cuda()
            rs.radix_sort(y, digits=d, ctx=c)
            c.check()
            assert np.array_equal(y.cpu().numpy(), orc.sort_parallel(u, lay, 8))
    c.close()


@pytest.mark.parametrize("t", ["u32", "(u64,u64)", "i16", "f64"])
@pytest.mark.parametrize("over", [0, 1])
def test_middle_size_bucket_capacity_edge(rs, torch, ctx, orc, t, over):
    """One top-digit bucket holds exactly what a workgroup of the bucket kernel takes (split accepted) or one element
    more (split refused on the device, LSD passes): same bytes as the oracle either way."""
    d = _digits(rs, t)
    es, ko, kb, _kind = util.TYPES[t]
    lay = orc.Layout(*util.TYPES[t])
    cap = 1024 * BUCKET_KEYS[es]
    n = cap + over + 60000
    raw = util.make_input(t, n, "uniform", seed=99 + over).reshape(n, es)
    rng = np.random.default_rng(5 + over)
    top = raw[:, ko + kb - 1]
    top[:] = rng.integers(0, 255, size=n, dtype=np.uint8)  # 0 .. 254
    top[top == 0x47] = 0x48
    pos = rng.choice(n, size=cap + over, replace=False)
    top[pos] = 0x47                                          # exactly cap (+ over) elements in bucket 0x47 (raw top byte)
    raw = raw.reshape(-1)
    assert np.array_equal(_gpu_sort(rs, torch, ctx, raw, d), orc.sort_parallel(raw, lay, 8)), (t, over)


@pytest.mark.parametrize("t", ["u16", "i16"])
def test_two_byte_counting_path(rs, torch, ctx, orc, t):
    """u16 / i16 arrays of at least 2^23 elements are sorted by counting (65536 LDS counters per workgroup as 16-bit halves,
    overflow parked in a global table, runs written from the bin totals): every distribution -- `equal` and `two` drive
    the counters over 0x8000 --, a size that is not a multiple of the 16-byte packs, and a 2-byte-aligned start."""
    d = _digits(rs, t)
    lay = orc.Layout(*util.TYPES[t])
    for i, dist in enumerate(util.DISTS):
        n = (1 << 23) + (0 if i % 2 else 1237)
        raw = util.make_input(t, n, dist, seed=70 + i)
        assert np.array_equal(_gpu_sort(rs, torch, ctx, raw, d), orc.sort_parallel(raw, lay, 8)), (t, n, dist)
    n = (1 << 23) + 5
    raw = util.make_input(t, n, "zipf", seed=91)
    buf = torch.zeros(2 * (n + 8), dtype=torch.uint8, device="cuda")
    x = buf[6:6 + 2 * n]  # 2-byte aligned, not 16
    x.copy_(torch.from_numpy(raw.copy()))
    rs.radix_sort(x, digits=d, ctx=ctx)
    ctx.check()
    assert np.array_equal(x.cpu().numpy(), orc.sort_parallel(raw, lay, 8)), (t, "odd offset")


@pytest.mark.parametrize("t,n", [("u8", 3000001), ("u16", 100003), ("u16", 5000001), ("u32", 3000001), ("u32", 300001), ("(u8,u8)", 777777)])
def test_sort_at_odd_element_offset(rs, torch, ctx, orc, t, n):
    """the C-ABI only asks for element alignment (the multi-GPU drivers pass interior pointers): a slice that
    starts one element into a 16-byte aligned buffer, data and scratch alike."""
    d = _digits(rs, t)
    lay = orc.Layout(*util.TYPES[t])
    es = d.elem_bytes
    raw = util.make_input(t, n, "uniform", seed=17)
    buf = torch.zeros(es * (n + 4), dtype=torch.uint8, device="cuda")
    tmp = torch.zeros(es * (n + 4), dtype=torch.uint8, device="cuda")
    x = buf[es:es * (n + 1)]
    x.copy_(torch.from_numpy(raw.copy()))
    rs.radix_sort(x, digits=d, ctx=ctx, tmp=tmp[es:es * (n + 1)])
    ctx.check()
    assert np.array_equal(x.cpu().numpy(), orc.sort_parallel(raw, lay, 8)), (t, n)
    assert buf[:es].cpu().numpy().sum() == 0 and buf[es * (n + 1):].cpu().numpy().sum() == 0  # nothing written outside the slice


@pytest.mark.parametrize("t", list(util.TYPES))
@pytest.mark.parametrize("dist", util.DISTS)
def test_distributions(rs, torch, ctx, orc, t, dist):
    d = _digits(rs, t)
    lay = orc.Layout(*util.TYPES[t])
    for n in (777, 50021):
        raw = util.make_input(t, n, dist, seed=hash((t, dist)) & 0xFFFF)
        assert np.array_equal(_gpu_sort(rs, torch, ctx, raw, d), orc.sort_parallel(raw, lay, 3)), (t, dist, n)


def test_golden_fixtures(rs, torch, ctx):
    z = np.load(GOLDEN)
    keys = [k[3:] for k in z.files if k.startswith("in|")]
    for k in keys:
        t = k.split("|")[0]
        got = _gpu_sort(rs, torch, ctx, z["in|" + k], _digits(rs, t))
        assert np.array_equal(got, z["out|" + k]), k


@pytest.mark.parametrize("t", ["u8", "u16", "u32", "u64", "u128", "i32", "f32", "f64", "(u32,u32)", "(u64,u64)"])
def test_reference_test_size_1e6(rs, torch, ctx, orc, t):
    """tests.rs shape: 1e6 random elements per type, vs the oracle (not just sortedness)."""
    d = _digits(rs, t)
    lay = orc.Layout(*util.TYPES[t])
    raw = util.make_input(t, 10 ** 6, "uniform", seed=42)
    assert np.array_equal(_gpu_sort(rs, torch, ctx, raw, d), orc.sort_parallel(raw, lay, 8))


def test_config1_1m_u32(rs, torch, ctx, orc):
    """BASELINE.json configs[0]: 1M u32 uniform keys -- CPU reference path (oracle) vs HIP."""
    n = 1 << 20
    raw = util.make_input("u32", n, "uniform", seed=0x5EED0001)
    lay = orc.Layout(4, 0, 4, 0)
    exp = orc.sort_parallel(raw, lay, os.cpu_count() or 1)
    assert np.array_equal(_gpu_sort(rs, torch, ctx, raw, rs.PRIMITIVES["u32"]), exp)


def test_stability_heavy_duplicates(rs, torch, ctx, orc):
    """tuples with few distinct keys: payload (= original index) order must be preserved."""
    for t in ("(u32,u32)", "(u64,u64)", "(u8,u8)", "(pay32+u32)"):
        raw = util.make_input(t, 300007, "two", seed=9)
        lay = orc.Layout(*util.TYPES[t])
        assert np.array_equal(_gpu_sort(rs, torch, ctx, raw, _digits(rs, t)), orc.sort0(raw, lay)), t


def test_idempotent_and_tmp_reuse(rs, torch, ctx):
    raw = util.make_input("u64", 200000, "uniform", seed=5)
    x = torch.from_numpy(raw.copy()).cuda()
    tmp = torch.empty_like(x)
    rs.radix_sort(x, digits=rs.PRIMITIVES["u64"], tmp=tmp)
    once = x.clone()
    rs.radix_sort(x, digits=rs.PRIMITIVES["u64"], tmp=tmp)
    ctx.check()
    assert torch.equal(x, once)


def test_torch_dtypes_inferred(rs, torch, ctx):
    g = torch.Generator(device="cpu").manual_seed(3)
    for dt in (torch.int32, torch.int64, torch.float32, torch.float64, torch.int16, torch.int8, torch.uint8):
        if dt.is_floating_point:
            x = torch.randn(100001, generator=g, dtype=dt)
        else:
            info = torch.iinfo(dt)
            x = torch.randint(info.min, info.max, (100001,), generator=g, dtype=torch.int64).to(dt)
        y = x.cuda()
        rs.radix_sort(y)
        ctx.check()
        assert torch.equal(y.cpu(), torch.sort(x, stable=True).values), dt


def test_host_slice_drop_in(rs, torch, ctx, orc):
    """rsx_sort_host: the literal `&mut [T]` drop-in (H2D -> sort -> D2H)."""
    a = np.random.default_rng(1).integers(0, 2 ** 32, size=300001, dtype=np.uint32)
    exp = np.sort(a, kind="stable")
    rs.radix_sort(a)
    assert np.array_equal(a, exp)
    pair = np.dtype([("k", "<u8"), ("v", "<u8")])
    b = np.zeros(100001, dtype=pair)
    b["k"] = np.random.default_rng(2).integers(0, 50, size=b.size)
    b["v"] = np.arange(b.size)
    exp = b[np.argsort(b["k"], kind="stable")]
    rs.radix_sort(b)
    assert np.array_equal(b, exp)


def test_histogram_and_partition_pass(rs, torch, ctx, orc):
    """The per-pass building blocks used by the multi-GPU bucket exchange."""
    for t in ("u32", "(u64,u64)", "f64"):
        es, ko, kb, kind = util.TYPES[t]
        d = _digits(rs, t)
        lay = orc.Layout(es, ko, kb, kind)
        raw = util.make_input(t, 70001, "zipf", seed=4)
        src = torch.from_numpy(raw.copy()).cuda()
        dst = torch.empty_like(src)
        hist = torch.zeros(256, dtype=torch.int64, device="cuda")
        for digit in range(kb):
            exp, exp_hist = orc.partition_pass(raw, lay, digit)
            ctx.histogram_device(src.data_ptr(), 70001, d, digit, hist.data_ptr())
            ctx.check()
            assert np.array_equal(hist.cpu().numpy().astype(np.uint64), exp_hist), (t, digit)
            hist.zero_()
            ctx.partition_device(src.data_ptr(), dst.data_ptr(), 70001, d, digit, hist.data_ptr())
            ctx.check()
            assert np.array_equal(dst.cpu().numpy(), exp), (t, digit)
            assert np.array_equal(hist.cpu().numpy().astype(np.uint64), exp_hist), (t, digit)


def test_segmented_copy(rs, torch, ctx):
    src = torch.arange(10000, dtype=torch.int64, device="cuda")
    dst = torch.full((10000,), -1, dtype=torch.int64, device="cuda")
    so = torch.tensor([0, 5000, 100], dtype=torch.int64, device="cuda")
    do = torch.tensor([7000, 0, 6000], dtype=torch.int64, device="cuda")
    ln = torch.tensor([3000, 5000, 0], dtype=torch.int64, device="cuda")
    ctx.segmented_copy_device(src.data_ptr(), dst.data_ptr(), 8, so.data_ptr(), do.data_ptr(), ln.data_ptr(), 3)
    ctx.check()
    exp = torch.full((10000,), -1, dtype=torch.int64)
    exp[7000:10000] = torch.arange(0, 3000)
    exp[0:5000] = torch.arange(5000, 10000)
    assert torch.equal(dst.cpu(), exp)


def test_errors(rs, torch, ctx):
    x = torch.zeros(1024, dtype=torch.uint8, device="cuda")
    with pytest.raises(rs.RsxError):  # 3-byte elements: no kernel
        rs.radix_sort(x[:1023], digits=rs.RadixDigits(3, 0, 2, 0))
    with pytest.raises(rs.RsxError):  # key outside the element
        rs.radix_sort(x, digits=rs.RadixDigits(4, 2, 4, 0))
    with pytest.raises(rs.RsxError):  # misaligned device pointer
        rs.radix_sort(x[1:1021], digits=rs.PRIMITIVES["u32"])


# ---- full-size configurations (BASELINE.json configs[1], configs[2]) through size-independent
# ---- properties: sortedness, multiset checksum, stability, all computed on the device.
def _full_size(rs, torch, ctx, t, n, gen, param=0.0, seed=0x5EED0002):
    d = _digits(rs, t)
    x = torch.empty(n * d.elem_bytes, dtype=torch.uint8, device="cuda")
    tmp = torch.empty_like(x)
    out = torch.zeros(3, dtype=torch.int64, device="cuda")
    ctx.generate_device(x.data_ptr(), n, d, gen, seed, param)
    ctx.verify_device(x.data_ptr(), n, d, out.data_ptr())
    before = out.cpu().numpy().astype(np.uint64)
    rs.radix_sort(x, digits=d, tmp=tmp)
    ctx.check()
    ctx.verify_device(x.data_ptr(), n, d, out.data_ptr())
    after = out.cpu().numpy().astype(np.uint64)
    assert after[0] == 0, f"{after[0]} descents after sort"
    assert after[1] == before[1], "multiset checksum changed"
    assert after[2] == 0, f"{after[2]} stability violations"
    if gen == rs.GEN_UNIFORM and n > 1000:
        assert before[0] > 0  # the input really was unsorted
    del x, tmp
    torch.cuda.empty_cache()


def test_config2_256m_u32(rs, torch, ctx):
    _full_size(rs, torch, ctx, "u32", 1 << 28, rs.GEN_UNIFORM)


def test_config3_1b_u64(rs, torch, ctx):
    _full_size(rs, torch, ctx, "u64", 1 << 30, rs.GEN_UNIFORM, seed=0x5EED0003)


def test_target_1b_u32(rs, torch, ctx):
    _full_size(rs, torch, ctx, "u32", 1 << 30, rs.GEN_UNIFORM, seed=0x5EED0006)


def test_pairs_zipf_stability_full_slice(rs, torch, ctx):
    """configs[4] per-GPU slice: 2^27 (u64 key, u64 payload=index) Zipf-skewed pairs."""
    _full_size(rs, torch, ctx, "(u64,u64)", 1 << 27, rs.GEN_ZIPF, param=1.0, seed=0x5EED0005)


def test_skewed_and_degenerate_large(rs, torch, ctx):
    for gen, param in ((rs.GEN_STEP, 16.0), (rs.GEN_CONSTANT, 7.0), (rs.GEN_SORTED, 0.0), (rs.GEN_REVERSED, 0.0),
                       (rs.GEN_ZIPF, 1.0)):
        _full_size(rs, torch, ctx, "(u32,u32)", (1 << 24) + 12345, gen, param)


def test_more_than_2pow30_elements(rs, torch, ctx):
    """n > 2^30: 64-bit element indices everywhere; all-equal keys push one digit's bucket past 2^30."""
    n = (1 << 30) + (1 << 20) + 77
    _full_size(rs, torch, ctx, "u8", n, rs.GEN_UNIFORM)
    # all-equal u8 key + 7-byte payload (= original index, wide enough not to wrap)
    _full_size(rs, torch, ctx, "(u8,[u8;7])", n, rs.GEN_CONSTANT, param=3.0)


def test_64bit_status_words_single_region(rs, torch):
    """Regions longer than 2^30 elements switch the look-back words to 64 bit.  With the default
    8-16 regions that takes n > 2^33; one region (one chain over everything, RSX_OPT_MAX_REGIONS = 1)
    reaches it at n > 2^30."""
    c = rs.Context(torch.cuda.current_device())
    c.set_option(rs.OPT_MAX_REGIONS, 1)
    n = (1 << 30) + (1 << 20) + 77
    for name, d, gen, param in (("u8", rs.PRIMITIVES["u8"], rs.GEN_UNIFORM, 0.0),
                                ("(u8,[u8;7])", rs.RadixDigits(8, 0, 1, 0), rs.GEN_CONSTANT, 3.0)):
        x = torch.empty(n * d.elem_bytes, dtype=torch.uint8, device="cuda")
        tmp = torch.empty_like(x)
        out = torch.zeros(3, dtype=torch.int64, device="cuda")
        c.generate_device(x.data_ptr(), n, d, gen, 7, param)
        c.verify_device(x.data_ptr(), n, d, out.data_ptr())
        torch.cuda.synchronize()
        before = out[1].item()
        rs.radix_sort(x, digits=d, tmp=tmp, ctx=c)
        c.check()
        c.verify_device(x.data_ptr(), n, d, out.data_ptr())
        torch.cuda.synchronize()
        assert out[0].item() == 0 and out[1].item() == before and out[2].item() == 0, (name, out.tolist())
        del x, tmp
        torch.cuda.empty_cache()
    c.close()


def test_verify_detects_errors(rs, torch, ctx):
    """The checker itself: unsorted data and instability must be reported."""
    d = _digits(rs, "(u32,u32)")
    n = 100000
    x = torch.empty(n * 8, dtype=torch.uint8, device="cuda")
    out = torch.zeros(3, dtype=torch.int64, device="cuda")
    ctx.generate_device(x.data_ptr(), n, d, rs.GEN_REVERSED, 1)
    ctx.verify_device(x.data_ptr(), n, d, out.data_ptr())
    assert out[0].item() == n - 1
    ctx.generate_device(x.data_ptr(), n, d, rs.GEN_CONSTANT, 1, 5.0)
    v = x.view(torch.int32).view(n, 2)
    v[:, 1] = torch.arange(n - 1, -1, -1, dtype=torch.int32, device="cuda")  # payload descending
    ctx.verify_device(x.data_ptr(), n, d, out.data_ptr())
    assert out[0].item() == 0 and out[2].item() == n - 1


ALT_PATHS = [("OPT_TILE_SCHEDULE", 1, "ticketed tiles instead of the static roll-call assignment"),
             ("OPT_RANKING", 1, "ranks by ballots only"),
             ("OPT_RANKING", 2, "ranks by returned LDS atomics whatever the skew"),
             ("OPT_STATUS_SCOPE", 1, "agent-scope status stores even on verified single-XCD chains"),
             ("OPT_XCD_MAJOR", 0, "no XCD-major workgroup numbering"),
             ("OPT_BYTE_COUNTING", 0, "one-byte elements through the general pass instead of the counting path"),
             ("OPT_MAX_REGIONS", 1, "one look-back chain over all tiles"),
             ("OPT_MAX_REGIONS", 32, "32 look-back chains"),
             ("OPT_HOT_LANES", 2, "every tile treated as skewed"),
             ("OPT_SMALL_SORT", 0, "arrays of at most one tile through the general path"),
             ("OPT_MID_SORT", 0, "middle sizes by LSD passes only (no bucket split)"),
             ("OPT_MID_SORT", 2, "middle sizes always split by the top digit (skewed inputs: oversized buckets through memory)"),
             ("OPT_MID_SORT", 3, "middle sizes always by LSD passes, top digit counted for the forecast"),
             ("OPT_WIDE_SORT", 2, "wide keys always by the 16-bit bucket hybrid"),
             ("OPT_WIDE_SORT", 0, "wide keys never by the 16-bit bucket hybrid"),
             ("OPT_BUCKET_SKIP", 0, "bucket kernels run every LDS pass (no skipped digits, no mending)"),
             ("OPT_BUCKET_GROUP", 0, "the hybrid sorts small 16-bit buckets one by one (no groups)")]


@pytest.mark.parametrize("opt,value,what", ALT_PATHS)
def test_alternative_kernel_paths_match(rs, torch, orc, opt, value, what):
    """Every fallback / alternative path of the sweep kernel (rsx_ctx_set_option) must give the same
    bytes as the oracle: these are the paths the library falls back to by itself when a device
    self-test fails or the grid is not co-resident."""
    c = rs.Context(torch.cuda.current_device())
    c.set_option(getattr(rs, opt), value)
    for t, n, dist in (("u32", 3000001, "uniform"), ("(u64,u64)", 700001, "zipf"), ("f32", 1500000, "uniform"),
                       ("u8", 5000000, "uniform"), ("i8", 3000001, "zipf"), ("(u8,[u8;7])", 2000003, "two"),
                       ("(u32,u32)", 1000001, "step16"), ("(i16,u16)", 1234567, "equal"), ("u64", 2000001, "lowbyte"),
                       ("(u32,[u8;8])", 500009, "zipf"),
                       # at most one tile: the one-launch kernel (or, with OPT_SMALL_SORT = 0, the general path on a tiny input)
                       ("u32", 1000, "uniform"), ("i64", 7168, "zipf"), ("(u64,u64)", 2560, "equal"), ("f32", 14336, "uniform"),
                       ("(u8,[u8;7])", 777, "two"), ("u16", 3, "uniform"), ("(u128,u128)", 1535, "lowbyte")):
        d = _digits(rs, t)
        raw = util.make_input(t, n, dist, seed=31)
        x = torch.from_numpy(raw.copy()).cuda()
        rs.radix_sort(x, digits=d, ctx=c)
        c.check()
        exp = orc.sort_parallel(raw, orc.Layout(*util.TYPES[t]), 8)
        assert np.array_equal(x.cpu().numpy(), exp), (what, t, dist)
    c.close()


def test_atomic_ranks_cross_checked_under_load(rs, torch, orc):
    """per-pass stability rests on the LDS applying same-address lanes of one ds_add_rtn in lane
    order.  rsx_lds_order_kernel tests that on an idle device; RSX_OPT_RANK_CHECK checks one round of every
    tile of REAL sweeps (two workgroups per CU hammering the LDS) against the ballot-derived ranks."""
    c = rs.Context(torch.cuda.current_device())
    c.set_option(rs.OPT_RANK_CHECK, 1)
    out = torch.zeros(3, dtype=torch.int64, device="cuda")
    for t, logn, gen, param in (("u32", 28, rs.GEN_UNIFORM, 0.0), ("u32", 26, rs.GEN_ZIPF, 1.0), ("u64", 27, rs.GEN_UNIFORM, 0.0),
                                ("(u64,u64)", 26, rs.GEN_STEP, 16.0), ("u16", 27, rs.GEN_UNIFORM, 0.0)):
        d = _digits(rs, t)
        n = 1 << logn
        x = torch.empty(n * d.elem_bytes, dtype=torch.uint8, device="cuda")
        tmp = torch.empty_like(x)
        for rep in range(3):
            c.generate_device(x.data_ptr(), n, d, gen, 99 + rep, param)
            rs.radix_sort(x, digits=d, tmp=tmp, ctx=c)
            c.check()  # raises if any cross-check failed
        c.verify_device(x.data_ptr(), n, d, out.data_ptr())
        v = out.cpu().tolist()
        assert v[0] == 0 and v[2] == 0, (t, v)
        del x, tmp
    c.close()


def test_options_reject_bad_values(rs, torch):
    c = rs.Context(torch.cuda.current_device())
    for opt, bad in ((rs.OPT_TILE_SCHEDULE, 2), (rs.OPT_RANKING, 3), (rs.OPT_MAX_REGIONS, 33), (rs.OPT_HOT_LANES, 1), (rs.OPT_MID_SORT, 4),
                     (rs.OPT_WIDE_SORT, 4), (rs.OPT_BUCKET_SKIP, 2), (rs.OPT_BUCKET_GROUP, 2), (99, 0)):
        with pytest.raises(rs.RsxError):
            c.set_option(opt, bad)
    c.close()


def test_device_self_tests_pass_on_gfx950(rs, torch, ctx):
    """The sweep ranks by returned LDS atomics only if rsx_lds_order_kernel passed on this device, and
    keeps single-XCD chains' status words in the L2 only if rsx_l2_probe_kernel did; on gfx950 both are
    expected to (otherwise the bench numbers are those of the fallback paths)."""
    assert ctx.get_info(rs.INFO_RANK_ATOMIC) == 1
    assert ctx.get_info(rs.INFO_L2_LOCAL) == 1


def test_many_sorts_one_context_and_graph_capture(rs, torch, ctx):
    """Workspace reuse across sizes/types, and stream-ordered operation: rsx_sort_device must be
    capturable in a HIP graph once the workspace is reserved (no allocation, no sync inside)."""
    g = torch.Generator(device="cpu").manual_seed(5)
    for n in (5, 70000, 9, 1 << 20, 12345, 3):
        x = torch.randint(0, 2 ** 31 - 1, (n,), generator=g, dtype=torch.int32)
        y = x.cuda()
        rs.radix_sort(y)
        ctx.check()
        assert torch.equal(y.cpu(), torch.sort(x).values)
    n = 1 << 20
    d = rs.PRIMITIVES["i32"]
    ctx.reserve(n, d)
    src = torch.randint(0, 2 ** 31 - 1, (n,), generator=g, dtype=torch.int32).cuda()
    work = torch.empty_like(src)
    tmp = torch.empty_like(src)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        work.copy_(src)
        ctx.sort_device(work.data_ptr(), tmp.data_ptr(), n, d, s.cuda_stream)  # warm-up outside capture
    s.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        work.copy_(src)
        ctx.sort_device(work.data_ptr(), tmp.data_ptr(), n, d, torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    ctx.check()
    assert torch.equal(work.cpu(), torch.sort(src.cpu()).values)


@pytest.mark.parametrize("t,n", [("u32", 300001), ("u32", 5000001), ("u64", 5000001), ("(u64,u64)", 3000001), ("f64", 4500001)])
def test_graph_replay_with_changing_inputs(rs, torch, orc, t, n):
    """A captured sort is replayed on different inputs: the middle-size path is captured as the bucket split (the
    forecast is taken at capture time) and a replay on a skewed input must still be right (oversized buckets through
    memory); the general path under capture uses the control block that it zeroes itself, replay after replay; wide keys
    above the middle sizes are captured as BOTH sequences of the hybrid, and every replay decides on the device which
    one runs (uniform: the hybrid, with the window where that input's keys differ; Zipf, equal keys: the LSD passes)."""
    c = rs.Context(torch.cuda.current_device())
    d = _digits(rs, t)
    lay = orc.Layout(*util.TYPES[t])
    c.reserve(n, d)
    inputs = [util.make_input(t, n, dist, seed=40 + i) for i, dist in enumerate(("uniform", "zipf", "uniform", "equal", "sorted"))]
    src = torch.from_numpy(inputs[0].copy()).cuda()
    work, tmp = torch.empty_like(src), torch.empty_like(src)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        work.copy_(src)
        c.sort_device(work.data_ptr(), tmp.data_ptr(), n, d, s.cuda_stream)  # warm-up outside capture (and the forecast's first report)
    s.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        work.copy_(src)
        c.sort_device(work.data_ptr(), tmp.data_ptr(), n, d, torch.cuda.current_stream().cuda_stream)
    for raw in inputs:
        src.copy_(torch.from_numpy(raw.copy()))
        graph.replay()
        torch.cuda.synchronize()
        c.check()
        assert np.array_equal(work.cpu().numpy(), orc.sort_parallel(raw, lay, 8))
    # an uncaptured sort between replays, then a replay again (the alternating control blocks are not the graph's)
    y = torch.from_numpy(inputs[1].copy()).cuda()
    rs.radix_sort(y, digits=d, ctx=c)
    c.check()
    assert np.array_equal(y.cpu().numpy(), orc.sort_parallel(inputs[1], lay, 8))
    src.copy_(torch.from_numpy(inputs[2].copy()))
    graph.replay()
    torch.cuda.synchronize()
    c.check()
    assert np.array_equal(work.cpu().numpy(), orc.sort_parallel(inputs[2], lay, 8))
    c.close()


# ---- rsx_sort_sharded: single-process multi-slice sort (slices share the one GPU of the test box) ----
SHARD_SPLITS = [
    [5000, 5000], [1, 9999], [0, 7000, 3000], [3333, 0, 3333, 3334], [12345], [100003, 50001, 70007],
    [2, 1, 0, 1], [8192, 8192, 8192, 8192, 8192, 8192, 8192, 8192],
]


@pytest.mark.parametrize("t", ["u8", "u32", "i32", "f32", "u64", "f64", "i128", "(u64,u64)", "(u8,[u8;7])",
                               "(pay32+u32)", "(u32,[u8;8])"])
@pytest.mark.parametrize("schedule", ["exchange_first", "sort_first"])
def test_sharded_single_process(rs, torch, orc, t, schedule):
    sched = rs.SHARD_EXCHANGE_FIRST if schedule == "exchange_first" else rs.SHARD_SORT_FIRST
    d = _digits(rs, t)
    lay = orc.Layout(*util.TYPES[t])
    es = util.TYPES[t][0]
    for si, split in enumerate(SHARD_SPLITS):
        for dist in ("uniform", "two", "zipf"):
            n = sum(split)
            raw = util.make_input(t, n, dist, seed=4242 + si)
            want = orc.sort_parallel(raw, lay, 4)
            flat = raw.reshape(-1).view(np.uint8)
            offs = np.concatenate(([0], np.cumsum(split))) * es
            slices = [torch.from_numpy(flat[offs[g]:offs[g + 1]].copy()).cuda() for g in range(len(split))]
            ctxs = [rs.Context(torch.cuda.current_device()) for _ in split]
            rs.radix_sort_sharded(slices, d, ctxs=ctxs, schedule=sched)
            got = np.concatenate([s.cpu().numpy() for s in slices]) if n else np.zeros(0, np.uint8)
            assert np.array_equal(got, want.reshape(-1).view(np.uint8)), (t, split, dist)
            for c in ctxs:
                c.close()


def test_sharded_argument_errors(rs, torch):
    d = rs.PRIMITIVES["u32"]
    c = rs.Context(torch.cuda.current_device())
    a = torch.zeros(16, dtype=torch.int32, device="cuda")
    b = torch.zeros(16, dtype=torch.int32, device="cuda")
    with pytest.raises(rs.RsxError):  # the same context twice
        rs.radix_sort_sharded([a, b], d, ctxs=[c, c])
    with pytest.raises(ValueError):
        rs.radix_sort_sharded([a, b], d, ctxs=[c])


def test_two_contexts_two_streams_concurrently(rs, torch, orc):
    """Two sorts in flight at once on one device (separate contexts and streams): their sweep kernels
    compete for the CUs, so the roll call of either may fail and fall back to ticketed tiles -- results
    must not care."""
    d = _digits(rs, "u32")
    lay = orc.Layout(*util.TYPES["u32"])
    n = 6_000_001
    raws = [util.make_input("u32", n, "uniform", seed=900 + i) for i in range(2)]
    want = [orc.sort_parallel(r, lay, 8) for r in raws]
    ctxs = [rs.Context(torch.cuda.current_device()) for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    xs = [torch.from_numpy(r.copy()).cuda() for r in raws]
    tmps = [torch.empty_like(x) for x in xs]
    torch.cuda.synchronize()
    for rep in range(6):
        for i in range(2):
            xs[i].copy_(torch.from_numpy(raws[i]).cuda(), non_blocking=False)
        torch.cuda.synchronize()
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                rs.radix_sort(xs[i], digits=d, tmp=tmps[i], ctx=ctxs[i])
        for i in range(2):
            streams[i].synchronize()
            ctxs[i].check(streams[i].cuda_stream)
            assert np.array_equal(xs[i].cpu().numpy(), want[i]), (rep, i)
    for c in ctxs:
        c.close()



# ---- BASELINE.json configs[3] and configs[4] at FULL size on the one GPU of the test box: one contiguous
# ---- buffer viewed as 8 equal slices, rsx_sort_sharded with 8 same-device contexts (both schedules), checked
# ---- (a) on device over the whole buffer (descents, multiset checksum, stability: slice boundaries included)
# ---- and (b) bit for bit against rsx_sort_device of a copy of the same elements.
def _equal_chunked(torch, a, b, chunk=1 << 30):
    for off in range(0, a.numel(), chunk):
        if not torch.equal(a[off:off + chunk], b[off:off + chunk]):
            return False
    return True


def _full_config_sharded(rs, torch, t, logn, gen, param, seed):
    d = _digits(rs, t)
    n, G = 1 << logn, 8
    es = d.elem_bytes
    dev = torch.cuda.current_device()
    c0 = rs.default_context(dev)
    x = torch.empty(n * es, dtype=torch.uint8, device="cuda")
    tmp = torch.empty_like(x)
    ref = torch.empty_like(x)
    out = torch.zeros(3, dtype=torch.int64, device="cuda")
    per = n // G
    ctxs = [rs.Context(dev) for _ in range(G)]
    for g in range(G):  # payload = GLOBAL index (index_base), keys from one global counter stream
        c0.generate_device(x.data_ptr() + g * per * es, per, d, gen, seed, param, g * per)
    c0.verify_device(x.data_ptr(), n, d, out.data_ptr())
    before = out.cpu().numpy().astype(np.uint64)
    assert before[0] > 0
    ref.copy_(x)
    rs.radix_sort(ref, digits=d, tmp=tmp, ctx=c0)  # the single-GPU sort of the same 2^logn elements
    c0.check()
    c0.verify_device(ref.data_ptr(), n, d, out.data_ptr())
    v = out.cpu().numpy().astype(np.uint64)
    assert v[0] == 0 and v[1] == before[1] and v[2] == 0, v
    keep = torch.empty_like(x)
    keep.copy_(x)
    for sched in (rs.SHARD_EXCHANGE_FIRST, rs.SHARD_SORT_FIRST):
        x.copy_(keep)
        slices = [x[g * per * es:(g + 1) * per * es] for g in range(G)]
        tmps = [tmp[g * per * es:(g + 1) * per * es] for g in range(G)]
        rs.radix_sort_sharded(slices, d, ctxs=ctxs, tmps=tmps, schedule=sched)
        c0.verify_device(x.data_ptr(), n, d, out.data_ptr())
        v = out.cpu().numpy().astype(np.uint64)
        assert v[0] == 0, f"{v[0]} descents (schedule {sched})"
        assert v[1] == before[1], "multiset checksum changed"
        assert v[2] == 0, f"{v[2]} stability violations"
        assert _equal_chunked(torch, x, ref), f"sharded result differs from the single-GPU sort (schedule {sched})"
    for c in ctxs:
        c.close()
    del x, tmp, ref, keep
    torch.cuda.empty_cache()


def test_config4_4b_u32_sharded_full_size(rs, torch):
    """configs[3]: 4B (2^32) u32 keys over 8 slices: 64-bit totals in the splitter search."""
    _full_config_sharded(rs, torch, "u32", 32, rs.GEN_UNIFORM, 0.0, 0x5EED0004)


def test_config5_1b_pairs_zipf_sharded_full_size(rs, torch):
    """configs[4]: 1B (2^30) (u64 key, u64 payload = global index) Zipf pairs over 8 slices: skew + stability."""
    _full_config_sharded(rs, torch, "(u64,u64)", 30, rs.GEN_ZIPF, 1.0, 0x5EED0005)


@pytest.mark.parametrize("t,logn", [("u64", 28), ("(u64,u64)", 27)])
def test_uniform_wide_keys_sharded(rs, torch, t, logn):
    """Uniform 8-byte keys over 8 slices: every slice sorts VALUE RANGES (three fixed top bits and more), which take the
    wide-key hybrid with its window below those bits; bit for bit the single sort (itself a hybrid sort of the whole)."""
    _full_config_sharded(rs, torch, t, logn, rs.GEN_UNIFORM, 0.0, 0x5EED0007)


def test_one_context_two_streams(rs, torch, orc):
    """one context used from two streams -- the second stream's sort must wait (on the device)
    for the first one's: they share the context's count matrices, tickets and status words."""
    d = _digits(rs, "u32")
    lay = orc.Layout(*util.TYPES["u32"])
    n = 5_000_003
    raws = [util.make_input("u32", n, "uniform", seed=700 + i) for i in range(2)]
    want = [orc.sort_parallel(r, lay, 8) for r in raws]
    c = rs.Context(torch.cuda.current_device())
    streams = [torch.cuda.Stream() for _ in range(2)]
    xs = [torch.from_numpy(r.copy()).cuda() for r in raws]
    tmps = [torch.empty_like(x) for x in xs]
    for rep in range(5):
        for i in range(2):
            xs[i].copy_(torch.from_numpy(raws[i]).cuda())
        torch.cuda.synchronize()
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                rs.radix_sort(xs[i], digits=d, tmp=tmps[i], ctx=c)
        for i in range(2):
            c.check(streams[i].cuda_stream)
            assert np.array_equal(xs[i].cpu().numpy(), want[i]), (rep, i)
    c.close()


def test_unreserved_sort_under_capture_reports_workspace(rs, torch):
    """A sort that would have to allocate while its stream is being captured returns RSX_ERR_WORKSPACE
    instead of calling hipMalloc mid-capture."""
    c = rs.Context(torch.cuda.current_device())
    d = rs.PRIMITIVES["u32"]
    n = 1 << 16
    x = torch.randint(0, 2 ** 31 - 1, (n,), dtype=torch.int32, device="cuda")
    tmp = torch.empty_like(x)
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    err = None
    with torch.cuda.stream(s):
        torch.cuda.synchronize()
        g.capture_begin()
        try:
            c.sort_device(x.data_ptr(), tmp.data_ptr(), n, d, torch.cuda.current_stream().cuda_stream)
        except rs.RsxError as e:
            err = e
        g.capture_end()
    assert err is not None and err.status == -6, err
    c.reserve(n, d)  # now it captures
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        g2.capture_begin()
        c.sort_device(x.data_ptr(), tmp.data_ptr(), n, d, torch.cuda.current_stream().cuda_stream)
        g2.capture_end()
    g2.replay()
    torch.cuda.synchronize()
    c.check()
    assert bool((x[1:] >= x[:-1]).all())
    c.close()


def test_two_streams_contention_costs_little(rs, torch):
    """two sorts in flight on two streams (two contexts) may lose the start-up roll call
    (the grids are not co-resident) -- that must cost a bounded wait once per sort, not a pass: the pair may
    not take more than 1.5x the two sorts run one after the other."""
    d = rs.PRIMITIVES["u32"]
    n = 1 << 26
    dev = torch.cuda.current_device()
    ctxs = [rs.Context(dev) for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    src = [torch.empty(n * 4, dtype=torch.uint8, device="cuda") for _ in range(2)]
    xs = [torch.empty_like(a) for a in src]
    tmps = [torch.empty_like(a) for a in src]
    for i in range(2):
        ctxs[i].reserve(n, d)
        ctxs[i].generate_device(src[i].data_ptr(), n, d, rs.GEN_UNIFORM, 11 + i)
    torch.cuda.synchronize()

    def run(concurrent):
        for i in range(2):
            xs[i].copy_(src[i])
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for i in range(2):
            st = streams[i] if concurrent else torch.cuda.current_stream()
            st.wait_event(ev0)
            with torch.cuda.stream(st):
                rs.radix_sort(xs[i], digits=d, tmp=tmps[i], ctx=ctxs[i])
        for i in range(2):
            if concurrent:
                torch.cuda.current_stream().wait_stream(streams[i])
        ev1.record()
        torch.cuda.synchronize()
        return ev0.elapsed_time(ev1)

    for _ in range(2):
        run(False), run(True)
    serial = sorted(run(False) for _ in range(7))[3]
    both = sorted(run(True) for _ in range(7))[3]
    for c in ctxs:
        c.check()
        c.close()
    assert both <= 1.5 * serial, (both, serial)
