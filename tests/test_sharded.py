"""Multi-rank bucket exchange (radix_sort_amd/sharded.py).

CPU (gloo, world_size 2 and 3): the exchange logic -- gathered histograms, digit-major /
rank-minor scan, all-to-all-v splits, segmented placement -- with the two LOCAL steps
supplied by a test-only CPU stand-in built on the oracle (the product backend is HIP).
Result must be bit-identical to the oracle's sort of the concatenated slices.
"""
import os
import socket
import sys

import numpy as np
import pytest

import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class OracleBackend:
    """TEST-ONLY stand-in for HipBackend: local partition pass via the oracle, numpy segmented copy."""

    def __init__(self):
        import torch
        from oracle import oracle
        self.torch, self.orc = torch, oracle

    def empty_bytes(self, nbytes):
        return self.torch.empty(nbytes, dtype=self.torch.uint8)

    def zeros_u64(self, n):
        return self.torch.zeros(n, dtype=self.torch.int64)

    def partition(self, src, dst, n, d, digit, hist):
        lay = self.orc.Layout(d.elem_bytes, d.key_offset, d.key_bytes, d.key_kind)
        out, h = self.orc.partition_pass(src.numpy()[: n * d.elem_bytes].copy(), lay, digit)
        dst.numpy()[: n * d.elem_bytes] = out
        hist.numpy()[:] = h.astype(np.int64)

    def partition_count(self, src, n, d, digit, nsub):
        lay = self.orc.Layout(d.elem_bytes, d.key_offset, d.key_bytes, d.key_kind)
        out = np.zeros((nsub, 256), dtype=np.int64)
        for k in range(nsub):
            a, b = (n * k) // nsub, (n * (k + 1)) // nsub
            if b > a:
                out[k] = self.orc.partition_pass(src.numpy()[a * d.elem_bytes:b * d.elem_bytes].copy(), lay, digit)[1].astype(np.int64)
        return out

    def partition_scatter(self, src, dst, n, d, digit, nsub, k):
        lay = self.orc.Layout(d.elem_bytes, d.key_offset, d.key_bytes, d.key_kind)
        a, b = (n * k) // nsub, (n * (k + 1)) // nsub
        if b > a:
            dst.numpy()[a * d.elem_bytes:b * d.elem_bytes] = self.orc.partition_pass(src.numpy()[a * d.elem_bytes:b * d.elem_bytes].copy(), lay, digit)[0]

    def segmented_copy(self, src, dst, elem_bytes, src_off, dst_off, length, nseg):
        s, t = src.numpy(), dst.numpy()
        for so, do, ln in zip(src_off.tolist(), dst_off.tolist(), length.tolist()):
            t[do * elem_bytes:(do + ln) * elem_bytes] = s[so * elem_bytes:(so + ln) * elem_bytes]

    def to_device_i64(self, a):
        return self.torch.from_numpy(a.astype(np.int64))

    def sort(self, x, tmp, n, d):
        lay = self.orc.Layout(d.elem_bytes, d.key_offset, d.key_bytes, d.key_kind)
        x.numpy()[: n * d.elem_bytes] = self.orc.sort_parallel(x.numpy()[: n * d.elem_bytes].copy(), lay, 2)

    def bounds(self, x, n, d, q_lo, q_hi):
        lay = self.orc.Layout(d.elem_bytes, d.key_offset, d.key_bytes, d.key_kind)
        cols = self.orc.numpy_mapped_key_columns(x.numpy()[: n * d.elem_bytes], lay)
        keys = [int.from_bytes(bytes(r), "little") for r in cols]  # sorted ascending (python ints: 128-bit safe)
        import bisect
        qs = [int(l) | (int(h) << 64) for l, h in zip(q_lo, q_hi)]
        less = np.array([bisect.bisect_left(keys, q) for q in qs], dtype=np.int64)
        leq = np.array([bisect.bisect_right(keys, q) for q in qs], dtype=np.int64)
        return less, leq

    def bounds_ranges(self, x, n, d, q_lo, q_hi, beg, end):
        less = np.zeros(len(q_lo), dtype=np.int64)
        leq = np.zeros(len(q_lo), dtype=np.int64)
        es = d.elem_bytes
        cache = {}
        for i, (b, e) in enumerate(zip(beg.tolist(), end.tolist())):
            if (b, e) not in cache:
                cache[(b, e)] = x[int(b) * es:int(e) * es]
            l, q = self.bounds(cache[(b, e)], int(e - b), d, q_lo[i:i + 1], q_hi[i:i + 1])
            less[i], leq[i] = l[0], q[0]
        return less, leq

    def finish(self):
        pass


def _worker(rank, world, port, tname, dist_name, sizes, outdir, schedule="per-pass"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import radix_sort_amd as rs
    from radix_sort_amd.sharded import ShardedRadixSort
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = rs.RadixDigits(*util.TYPES[tname])
        total = sum(sizes)
        full = util.make_input(tname, total, dist_name, seed=77)
        off = sum(sizes[:rank])
        mine = full[off * d.elem_bytes:(off + sizes[rank]) * d.elem_bytes].copy()
        x = torch.from_numpy(mine)
        sorter = ShardedRadixSort(backend=OracleBackend())
        {"per-pass": sorter.sort, "one": sorter.sort_one_exchange, "first": sorter.sort_exchange_first,
         "pipelined": lambda *a, **k: sorter.sort_exchange_first(*a, chunks=3, **k),
         "overlapped": lambda *a, **k: sorter.sort_exchange_first(*a, chunks=3, sub_ranges=3, **k),
         "overlapped2": lambda *a, **k: sorter.sort_exchange_first(*a, chunks=1, sub_ranges=2, **k)}[schedule](x, d, n_per_rank=list(sizes))
        if schedule == "overlapped":  # the donated form hands back another buffer
            y = torch.from_numpy(mine.copy())
            r = sorter.sort_exchange_first(y, d, n_per_rank=list(sizes), chunks=2, sub_ranges=4, donate=True)
            assert np.array_equal(r.numpy(), x.numpy()), "donated result differs"
        np.save(os.path.join(outdir, f"out{rank}.npy"), x.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,tname,dist_name,sizes", [
    (2, "u32", "uniform", (5000, 5000)),
    (2, "(u64,u64)", "zipf", (4097, 4097)),
    (2, "f32", "uniform", (3000, 1234)),       # ragged slices
    (3, "(u32,u32)", "two", (1000, 0, 2500)),  # an empty slice, heavy duplicates
    (2, "i16", "highbyte", (2048, 2049)),
])
def test_sharded_gloo_matches_single_sort(orc, tmp_path, world, tname, dist_name, sizes):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, tname, dist_name, sizes, str(tmp_path)), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f"out{r}.npy") for r in range(world)])
    lay = orc.Layout(*util.TYPES[tname])
    full = util.make_input(tname, sum(sizes), dist_name, seed=77)
    assert np.array_equal(got, orc.sort_parallel(full, lay, 3))


@pytest.mark.parametrize("world,tname,dist_name,sizes", [
    (2, "u32", "uniform", (5000, 5000)),
    (2, "(u64,u64)", "zipf", (4097, 4097)),
    (2, "f32", "uniform", (3000, 1234)),
    (3, "(u32,u32)", "two", (1000, 0, 2500)),   # heavy ties straddling both boundaries, an empty slice
    (3, "i16", "equal", (700, 900, 800)),       # every key equal: all splitting is by tie order
    (2, "f64", "uniform", (2048, 2049)),
    (2, "u128", "uniform", (1500, 1500)),       # wide keys: 128-bit splitters
    (3, "(u8,u8)", "uniform", (900, 1100, 1000)),  # one-digit keys: the top digit is the whole key
    (3, "u32", "sorted", (2000, 1, 3000)),
])
@pytest.mark.parametrize("schedule", ["one", "first", "pipelined", "overlapped", "overlapped2"])
def test_one_exchange_gloo_matches_single_sort(orc, tmp_path, world, tname, dist_name, sizes, schedule):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, tname, dist_name, sizes, str(tmp_path), schedule), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f"out{r}.npy") for r in range(world)])
    lay = orc.Layout(*util.TYPES[tname])
    full = util.make_input(tname, sum(sizes), dist_name, seed=77)
    assert np.array_equal(got, orc.sort_parallel(full, lay, 3))


def test_exchange_plan_properties():
    """Host-only: split sizes are consistent across ranks and segments tile each output slice."""
    from radix_sort_amd.sharded import exchange_plan
    rng = np.random.default_rng(0)
    for G in (1, 2, 4, 8):
        n_per = rng.integers(0, 5000, size=G)
        H = np.zeros((G, 256), dtype=np.int64)
        for g in range(G):  # random histogram with the right total, some empty bins
            bins = rng.integers(0, 256, size=n_per[g])
            H[g] = np.bincount(bins[bins % 3 != 0] if n_per[g] else bins, minlength=256)
            H[g, 0] += n_per[g] - H[g].sum()
        plans = [exchange_plan(H, n_per, r) for r in range(G)]
        for r in range(G):
            send, recv, segs = plans[r]
            assert send.sum() == n_per[r] and recv.sum() == n_per[r]
            for h in range(G):
                assert send[h] == plans[h][1][r]  # what r sends to h is what h expects from r
            cover = np.zeros(n_per[r], dtype=np.int32)
            for so, do, ln in segs:
                cover[do:do + ln] += 1
            assert (cover == 1).all()
            # pieces sit in the receive buffer without gaps or overlap as well
            src_cover = np.zeros(n_per[r], dtype=np.int32)
            for so, do, ln in segs:
                src_cover[so:so + ln] += 1
            assert (src_cover == 1).all()


# ---- GPU: the real HIP local steps under the same exchange, 2 ranks sharing one MI355X ----------
def _gpu_worker(rank, world, port, tname, dist_name, sizes, outdir, schedule="per-pass"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import radix_sort_amd as rs
    from radix_sort_amd.sharded import ShardedRadixSort
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = rs.RadixDigits(*util.TYPES[tname])
        full = util.make_input(tname, sum(sizes), dist_name, seed=78)
        off = sum(sizes[:rank])
        x = torch.from_numpy(full[off * d.elem_bytes:(off + sizes[rank]) * d.elem_bytes].copy()).cuda()
        sorter = ShardedRadixSort()  # product backend: HIP through the C-ABI
        {"per-pass": sorter.sort, "one": sorter.sort_one_exchange, "first": sorter.sort_exchange_first,
         "pipelined": lambda *a, **k: sorter.sort_exchange_first(*a, chunks=4, **k),
         "overlapped": lambda *a, **k: sorter.sort_exchange_first(*a, chunks=3, sub_ranges=4, **k)}[schedule](x, d, n_per_rank=list(sizes))
        np.save(os.path.join(outdir, f"out{rank}.npy"), x.cpu().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("tname,dist_name,sizes", [
    ("u32", "uniform", (300000, 300000)),
    ("(u64,u64)", "zipf", (150001, 99999)),
    ("f64", "uniform", (70000, 1)),
])
def test_sharded_hip_two_ranks_one_gpu(orc, tmp_path, tname, dist_name, sizes):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_gpu_worker, args=(2, port, tname, dist_name, sizes, str(tmp_path)), nprocs=2, join=True)
    got = np.concatenate([np.load(tmp_path / f"out{r}.npy") for r in range(2)])
    lay = orc.Layout(*util.TYPES[tname])
    full = util.make_input(tname, sum(sizes), dist_name, seed=78)
    assert np.array_equal(got, orc.sort_parallel(full, lay, 4))


@pytest.mark.gpu
@pytest.mark.parametrize("tname,dist_name,sizes", [
    ("u32", "uniform", (300000, 300000)),
    ("(u64,u64)", "zipf", (150001, 99999)),
    ("f64", "uniform", (70000, 1)),
    ("(u32,u32)", "two", (200000, 123457)),
    ("i16", "equal", (65536, 70000)),
])
@pytest.mark.parametrize("schedule", ["one", "first", "pipelined", "overlapped"])
def test_one_exchange_hip_two_ranks_one_gpu(orc, tmp_path, tname, dist_name, sizes, schedule):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_gpu_worker, args=(2, port, tname, dist_name, sizes, str(tmp_path), schedule), nprocs=2, join=True)
    got = np.concatenate([np.load(tmp_path / f"out{r}.npy") for r in range(2)])
    lay = orc.Layout(*util.TYPES[tname])
    full = util.make_input(tname, sum(sizes), dist_name, seed=78)
    assert np.array_equal(got, orc.sort_parallel(full, lay, 4))


def _gpu_big_worker(rank, world, port, logn, schedule):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import radix_sort_amd as rs
    from radix_sort_amd.sharded import ShardedRadixSort
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = rs.tuple_of("u64", 8)
        n = 1 << logn
        ctx = rs.default_context(0)
        x = torch.empty(n * d.elem_bytes, dtype=torch.uint8, device="cuda")
        # configs[4]'s slice: Zipf(1) u64 keys, payload = GLOBAL index (rank * n + i): reveals instability across ranks too
        ctx.generate_device(x.data_ptr(), n, d, rs.GEN_ZIPF, 0x5EED0004, 1.0, rank * n, torch.cuda.current_stream().cuda_stream)
        sorter = ShardedRadixSort()
        before = sorter.checksum(x, d)
        if schedule == "per-pass":
            sorter.sort(x, d, [n] * world)
        elif schedule == "overlapped":
            x = sorter.sort_exchange_first(x, d, [n] * world, chunks=4, sub_ranges=4, donate=True)
        else:
            sorter.sort_exchange_first(x, d, [n] * world, chunks=4)
        sorter.verify(x, d, before)  # the cross-rank check bench.py --gpus N runs
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("schedule", ["pipelined", "overlapped", "per-pass"])
def test_torch_distributed_path_at_baseline_size(schedule):
    """the torch.distributed path at a BASELINE size under pytest: 2 gloo ranks on the one GPU over
    2 x 2^27 (u64,u64) Zipf pairs (configs[4]'s slice per GPU), payload = global index, through the exchange-first
    schedule (4 ranges; 4 sub-ranges x 4 ranges) and the per-pass schedule (the north star's), checked by
    ShardedRadixSort.verify: slices sorted and stable, multiset checksum all-reduced, rank-boundary keys in order."""
    import torch.multiprocessing as mp
    mp.spawn(_gpu_big_worker, args=(2, _free_port(), 27, schedule), nprocs=2, join=True)


@pytest.mark.gpu
def test_rccl_backend_world1():
    """The RCCL ("nccl") backend has never moved a byte between two GPUs here (one-GPU boxes).  With ONE rank it
    still runs every collective call of the three schedules and of bench.py's cross-rank check through RCCL:
    argument types, dtypes and split lists are accepted, results verified on the device."""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_world1.py")
    out = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=600)
    print(out.stdout[-2000:], out.stderr[-2000:])
    assert out.returncode == 0 and "RCCL WORLD1 OK" in out.stdout
