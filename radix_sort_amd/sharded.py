"""Multi-GPU LSD radix sort: one contiguous slice per rank, one bucket exchange per pass.

This is the reference's chunked scheme (src/radix_sort/mod.rs:90-168) with
"chunk per OS thread" replaced by "slice per GPU":

  per pass d (mod.rs:84):
    count    each rank's 256-bin histogram of digit d             (mod.rs:90-109)
    prefix   all-gather the G x 256 counts; every rank computes the same
             digit-major, RANK-minor exclusive scan                (mod.rs:110-120)
    scatter  local stable partition by digit d (HIP onesweep pass), then an
             all-to-all-v moves each run to the rank owning its global range, and
             a segmented copy places the received (digit, source-rank) runs in
             digit-major, rank-minor order                         (mod.rs:121-168)

One process per GPU; collectives go through torch.distributed (backend "nccl" is
RCCL over xGMI on ROCm; "gloo" on CPU is used by the world_size-2 tests).  The
output is bit-identical to the single-GPU sort of the concatenated slices.

The per-rank compute steps are behind a small backend interface; the product
backend is `HipBackend` (C-ABI calls into librsx.so).  Tests may inject a CPU
stand-in for the two local steps to exercise the exchange logic without a GPU.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np

from .api import Context, RadixDigits, default_context


class HipBackend:
    """Local steps on the GPU through the C-ABI (rsx_partition_device / rsx_segmented_copy_device)."""

    def __init__(self, ctx: Optional[Context] = None):
        import torch
        self.torch = torch
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.ctx = ctx or default_context(self.device.index)

    def empty_bytes(self, nbytes: int):
        return self.torch.empty(nbytes, dtype=self.torch.uint8, device=self.device)

    def zeros_u64(self, n: int):
        return self.torch.zeros(n, dtype=self.torch.int64, device=self.device)

    def partition(self, src, dst, n: int, d: RadixDigits, digit: int, hist):
        s = self.torch.cuda.current_stream().cuda_stream
        self.ctx.partition_device(src.data_ptr(), dst.data_ptr(), n, d, digit, hist.data_ptr(), s)

    def partition_count(self, src, n: int, d: RadixDigits, digit: int, nsub: int) -> np.ndarray:
        """Digit counts of every position sub-range of the slice (rsx_partition_count_device): numpy int64 [nsub, 256]."""
        hist = self.torch.zeros(nsub * 256, dtype=self.torch.int64, device=self.device)
        s = self.torch.cuda.current_stream().cuda_stream
        self.ctx.partition_count_device(src.data_ptr(), n, d, digit, nsub, hist.data_ptr(), s)
        return hist.cpu().numpy().reshape(nsub, 256)

    def partition_scatter(self, src, dst, n: int, d: RadixDigits, digit: int, nsub: int, k: int):
        """Stable partition of sub-range k by `digit` into the same positions of dst (rsx_partition_scatter_device)."""
        s = self.torch.cuda.current_stream().cuda_stream
        self.ctx.partition_scatter_device(src.data_ptr(), dst.data_ptr(), n, d, digit, nsub, k, s)

    def search_cuts_device(self, bb, nbb: int, d: RadixDigits, beg, end, pre_lo, pre_hi, rank_in, top_digit: int, all_reduce):
        """ShardedRadixSort._search_cuts with the key prefix kept on the device: per digit one count kernel, the ranks'
        all-reduce (`all_reduce(tensor)`, in place) and one pick kernel, all stream-ordered; ONE copy to the host at the end."""
        t = self.torch
        nb = len(rank_in)
        rng = np.stack([np.asarray(beg, dtype=np.uint64), np.asarray(end, dtype=np.uint64)], axis=1).reshape(-1)
        pre = np.stack([np.asarray(pre_lo, dtype=np.uint64), np.asarray(pre_hi, dtype=np.uint64)], axis=1).reshape(-1)
        host = np.concatenate([rng, pre, np.asarray(rank_in, dtype=np.int64).view(np.uint64)]).view(np.int64)
        dev = t.from_numpy(host.copy()).to(self.device)
        ranges, prefix, rank = dev[:2 * nb], dev[2 * nb:4 * nb], dev[4 * nb:]
        less = t.empty(nb * 256, dtype=t.int64, device=self.device)
        out = t.empty(2 * nb, dtype=t.int64, device=self.device)
        s = t.cuda.current_stream().cuda_stream
        for digit in range(top_digit - 1, -1, -1):
            self.ctx.splitter_count_device(bb.data_ptr(), nbb, d, ranges.data_ptr(), prefix.data_ptr(), nb, digit, less.data_ptr(), s)
            all_reduce(less)
            self.ctx.splitter_pick_device(less.data_ptr(), rank.data_ptr(), prefix.data_ptr(), nb, digit, s)
        self.ctx.bounds_ranges_device(bb.data_ptr(), nbb, d, prefix.data_ptr(), ranges.data_ptr(), nb, out.data_ptr(), s)
        o = out.cpu().numpy()
        return o[:nb].copy(), o[nb:].copy()

    def segmented_copy(self, src, dst, elem_bytes: int, src_off, dst_off, length, nseg: int):
        s = self.torch.cuda.current_stream().cuda_stream
        self.ctx.segmented_copy_device(src.data_ptr(), dst.data_ptr(), elem_bytes, src_off.data_ptr(),
                                       dst_off.data_ptr(), length.data_ptr(), nseg, s)

    def to_device_i64(self, a: np.ndarray):
        return self.torch.from_numpy(a.astype(np.int64)).to(self.device, non_blocking=False)

    def sort(self, x, tmp, n: int, d: RadixDigits):
        """Full local sort (rsx_sort_device), in place in x."""
        s = self.torch.cuda.current_stream().cuda_stream
        self.ctx.sort_device(x.data_ptr(), tmp.data_ptr(), n, d, s)

    def bounds(self, x, n: int, d: RadixDigits, q_lo: np.ndarray, q_hi: np.ndarray):
        """(less, less_or_equal) counts of the 128-bit mapped-key queries in the sorted slice x
        (rsx_bounds_device); numpy int64 arrays."""
        nq = len(q_lo)
        q = np.empty((nq, 2), dtype=np.uint64)
        q[:, 0], q[:, 1] = q_lo, q_hi
        dq = self.torch.from_numpy(q.view(np.int64).reshape(-1)).to(self.device)
        out = self.torch.empty(2 * nq, dtype=self.torch.int64, device=self.device)
        s = self.torch.cuda.current_stream().cuda_stream
        self.ctx.bounds_device(x.data_ptr(), n, d, dq.data_ptr(), nq, out.data_ptr(), s)
        o = out.cpu().numpy()
        return o[:nq].copy(), o[nq:].copy()

    def bounds_ranges(self, x, n: int, d: RadixDigits, q_lo: np.ndarray, q_hi: np.ndarray, beg: np.ndarray, end: np.ndarray):
        """The same for queries that each name their own sorted range [beg, end) of x (rsx_bounds_ranges_device):
        one launch and one round trip for all boundaries; counts are relative to the range's start."""
        nq = len(q_lo)
        q = np.empty((2 * nq, 2), dtype=np.uint64)
        q[:nq, 0], q[:nq, 1] = q_lo, q_hi
        q[nq:, 0], q[nq:, 1] = beg, end
        dq = self.torch.from_numpy(q.view(np.int64).reshape(-1)).to(self.device)
        out = self.torch.empty(2 * nq, dtype=self.torch.int64, device=self.device)
        s = self.torch.cuda.current_stream().cuda_stream
        self.ctx.bounds_ranges_device(x.data_ptr(), n, d, dq.data_ptr(), dq.data_ptr() + 16 * nq, nq, out.data_ptr(), s)
        o = out.cpu().numpy()
        return o[:nq].copy(), o[nq:].copy()

    def finish(self):
        self.ctx.check(self.torch.cuda.current_stream().cuda_stream)


def exchange_plan(H: np.ndarray, n_per_rank: np.ndarray, rank: int):
    """Pure host logic of one pass, identical on every rank.

    H[g][v]   = count of digit v on rank g (the gathered histograms)
    n_per_rank[g] = slice length of rank g (fixed across passes)
    Returns (send_counts[G], recv_counts[G], segments) for `rank`, where segments is an
    (nseg, 3) int64 array of (src_off, dst_off, len) in ELEMENTS: src_off indexes the
    receive buffer (chunks ordered by source rank), dst_off the rank's output slice.

    The global position of run (v, g) is the digit-major, rank-minor exclusive scan of H
    (mod.rs:110-120 with chunk == rank).  Because that position is monotone in v for a
    fixed source g, what g sends to a given destination is ONE contiguous range of its
    locally partitioned slice.
    """
    H = np.asarray(H, dtype=np.int64)
    G = H.shape[0]
    n_per_rank = np.asarray(n_per_rank, dtype=np.int64)
    bounds = np.concatenate(([0], np.cumsum(n_per_rank)))  # rank h owns [bounds[h], bounds[h+1])
    flat = H.T.reshape(-1)  # order (v, g): digit-major, rank-minor
    start = (np.cumsum(flat) - flat).reshape(256, G)  # S[v][g]
    length = H.T  # len[v][g]
    end = start + length
    # piece of run (v, g) that lands on destination h: overlap with [bounds[h], bounds[h+1])
    lo = np.maximum(start[None, :, :], bounds[:-1, None, None])  # [h][v][g]
    hi = np.minimum(end[None, :, :], bounds[1:, None, None])
    piece = np.clip(hi - lo, 0, None)  # [h][v][g]
    sent = piece.sum(axis=1)  # [h][g] elements g -> h
    send_counts = sent[:, rank].copy()  # what `rank` sends to each h
    recv_counts = sent[rank, :].copy()  # what `rank` receives from each g
    # receive buffer: chunk from g starts at recv_base[g]; inside it pieces are ordered by v
    recv_base = np.cumsum(recv_counts) - recv_counts
    mine = piece[rank]  # [v][g]
    within = np.cumsum(mine, axis=0) - mine  # offset of piece v inside g's chunk
    src_off = recv_base[None, :] + within
    dst_off = lo[rank] - bounds[rank]
    sel = mine > 0
    segs = np.stack([src_off[sel], dst_off[sel], mine[sel]], axis=1).astype(np.int64)
    return send_counts, recv_counts, segs


class ShardedRadixSort:
    """`radix_sort` over a slice-per-rank array.  All ranks call `sort` collectively.

    Three schedules with the same (unique) result:
      * `sort`                one bucket exchange per pass -- the reference's loop with chunk == rank;
      * `sort_one_exchange`   sort locally, find the exact splitters of the destination slices by a
                              256-way search over the key digits (one small all-reduce per digit),
                              exchange ONCE, sort locally again (the received chunks arrive ordered by
                              source rank, so a stable local sort reproduces the global stable order);
      * `sort_exchange_first` ONE stable partition pass by the most significant digit, the G x 256
                              counts laid out globally (mod.rs:110-120 with chunk == rank); only the
                              buckets that a slice boundary falls into are sorted locally and cut
                              exactly; exchange ONCE; ONE local sort.  1 + D local passes instead of 2 D.
    xGMI moves (G-1)/G of every slice per exchange at a small fraction of HBM speed, so the number
    of exchanges first, then the local passes, decide multi-GPU throughput.
    """

    def __init__(self, group=None, backend=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = backend or HipBackend()
        self._bufs = {}
        # gloo moves host memory only: device slices are staged through the host for the two
        # collectives (test rigs: several ranks on one GPU).  RCCL ("nccl") runs device to device.
        self.host_staged = dist.get_backend(group) == "gloo"
        # the small collectives of the planning steps go through a group of their own: on RCCL a communicator is a
        # stream, and the data group's is full of queued exchange batches while the cuts are being searched
        self.ctl = dist.new_group(backend=dist.get_backend(group)) if group is None and self.world > 1 else group
        # RSX_SHARD_TIMING=1: rank 0 prints where a sort_exchange_first call spends its time (each mark
        # synchronises the device: diagnostics only, it serialises what the schedule overlaps)
        import os
        self.timing = os.environ.get("RSX_SHARD_TIMING", "0") not in ("", "0") and self.rank == 0
        self._marks = []
        self.last_branch = ""  # which exchange code ran in the last sort (bench.py records it)

    def _mark(self, label: str):
        if not self.timing:
            return
        import time
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        except ImportError:
            pass
        self._marks.append((label, time.perf_counter()))

    def _report(self):
        if self.timing and len(self._marks) > 1:
            t0 = self._marks[0][1]
            parts = [f"{b[0]} {1e3 * (b[1] - a[1]):.3f}" for a, b in zip(self._marks, self._marks[1:])]
            print(f"[rsx sharded] total {1e3 * (self._marks[-1][1] - t0):.3f} ms: " + " | ".join(parts), flush=True)
        self._marks = []

    def _buf(self, name: str, nbytes: int):
        b = self._bufs.get(name)
        if b is None or b.numel() < nbytes:
            b = self._bufs[name] = self.backend.empty_bytes(nbytes)
        return b[:nbytes]

    def _gather_counts(self, n_local: int, device) -> List[int]:
        import torch
        dev = torch.device("cpu") if self.host_staged else device
        t = torch.tensor([n_local], dtype=torch.int64, device=dev)
        allt = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(allt, t, group=self.group)
        return [int(a.item()) for a in allt]

    def _all_reduce_sum(self, a: np.ndarray, device, ctl: bool = False) -> np.ndarray:
        import torch
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64))
        if not self.host_staged:
            t = t.to(device)
        self.dist.all_reduce(t, group=self.ctl if ctl else self.group)
        return t.cpu().numpy()

    def sort_one_exchange(self, x, d: RadixDigits, n_per_rank: Optional[List[int]] = None):
        """Same contract as `sort`; one all-to-all instead of one per digit."""
        import torch
        dist, be = self.dist, self.backend
        es, G = d.elem_bytes, self.world
        n_local = x.numel() // es
        if n_per_rank is None:
            n_per_rank = self._gather_counts(n_local, x.device)
        n_per_rank = np.asarray(n_per_rank, dtype=np.int64)
        assert n_per_rank[self.rank] == n_local
        bounds = np.concatenate(([0], np.cumsum(n_per_rank)))
        tmp = self._buf("part", n_local * es)
        recv = self._buf("recv", n_local * es)
        # 1. local sort (stable)
        if n_local > 1:
            be.sort(x, tmp, n_local, d)
        # 2. exact splitters: for every interior boundary T_h the key K_h with
        #    global_less(K_h) <= T_h < global_less_or_equal(K_h), found digit by digit (256-way);
        #    keys are 128-bit (lo, hi) mapped keys, counted by binary search in every sorted slice
        targets = bounds[1:-1]  # G-1 boundaries
        nb = len(targets)
        pre_lo = np.zeros(nb, dtype=np.uint64)
        pre_hi = np.zeros(nb, dtype=np.uint64)
        j256 = np.arange(256, dtype=np.uint64)
        for digit in range(d.key_bytes - 1, -1, -1):
            lo = np.repeat(pre_lo[:, None], 256, axis=1)
            hi = np.repeat(pre_hi[:, None], 256, axis=1)
            if digit < 8:
                lo |= j256[None, :] << np.uint64(8 * digit)
            else:
                hi |= j256[None, :] << np.uint64(8 * (digit - 8))
            less, _ = be.bounds(x, n_local, d, lo.reshape(-1), hi.reshape(-1))
            gl = self._all_reduce_sum(less, x.device).reshape(nb, 256)
            # largest candidate whose global "less" count does not exceed the target
            j = ((gl <= targets[:, None]).sum(axis=1) - 1).astype(np.uint64)  # gl[:,0] counts keys < prefix: always <= target
            if digit < 8:
                pre_lo |= j << np.uint64(8 * digit)
            else:
                pre_hi |= j << np.uint64(8 * (digit - 8))
        less_me, leq_me = be.bounds(x, n_local, d, pre_lo, pre_hi) if nb else (np.zeros(0, np.int64), np.zeros(0, np.int64))
        mine = np.stack([less_me, leq_me]).astype(np.int64)  # [2][nb]
        allm = [torch.zeros(2 * nb, dtype=torch.int64) for _ in range(G)]
        src = torch.from_numpy(mine.reshape(-1))
        if self.host_staged:
            dist.all_gather(allm, src, group=self.group)
        else:
            allm = [a.to(x.device) for a in allm]
            dist.all_gather(allm, src.to(x.device), group=self.group)
        M = torch.stack([a.cpu() for a in allm]).numpy().reshape(G, 2, nb)
        less, eq = M[:, 0, :], M[:, 1, :] - M[:, 0, :]  # [G][nb]
        # ties on K_h are dealt out in rank order (stability: lower source rank first)
        need = targets - less.sum(axis=0)  # elements equal to K_h that go below boundary h
        before = np.cumsum(eq, axis=0) - eq
        take = np.clip(need[None, :] - before, 0, eq)
        split = np.concatenate([np.zeros((G, 1), np.int64), less + take, n_per_rank[:, None]], axis=1)  # [G][G+1]
        send_counts = np.diff(split[self.rank])  # to each destination
        recv_counts = split[:, self.rank + 1] - split[:, self.rank]  # from each source
        assert send_counts.sum() == n_local and recv_counts.sum() == n_local, (send_counts, recv_counts)
        # 3. the one exchange
        if self.host_staged and x.is_cuda:
            rc = torch.empty(n_local * es, dtype=torch.uint8)
            dist.all_to_all_single(rc, x.cpu(), output_split_sizes=(recv_counts * es).tolist(),
                                   input_split_sizes=(send_counts * es).tolist(), group=self.group)
            recv.copy_(rc)
        else:
            dist.all_to_all_single(recv, x, output_split_sizes=(recv_counts * es).tolist(),
                                   input_split_sizes=(send_counts * es).tolist(), group=self.group)
        # 4. chunks arrive ordered by source rank, each sorted: a stable sort merges them
        x.copy_(recv)
        if n_local > 1:
            be.sort(x, tmp, n_local, d)
        be.finish()

    def _all_gather_i64(self, a: np.ndarray, device, ctl: bool = False) -> np.ndarray:
        """[G, len(a)] int64: every rank's vector (ctl: on the control group)."""
        import torch
        grp = self.ctl if ctl else self.group
        src = torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64))
        if self.host_staged:
            out = [torch.zeros_like(src) for _ in range(self.world)]
            self.dist.all_gather(out, src, group=grp)
            return torch.stack(out).numpy()
        src = src.to(device)
        out = [torch.zeros_like(src) for _ in range(self.world)]
        self.dist.all_gather(out, src, group=grp)
        return torch.stack(out).cpu().numpy()

    def _exchange(self, send, recv, send_counts, recv_counts, es: int):
        import torch
        if self.host_staged and send.is_cuda:
            rc = torch.empty(recv.numel(), dtype=torch.uint8)
            self.dist.all_to_all_single(rc, send.cpu(), output_split_sizes=(recv_counts * es).tolist(),
                                        input_split_sizes=(send_counts * es).tolist(), group=self.group)
            recv.copy_(rc)
        else:
            self.dist.all_to_all_single(recv, send, output_split_sizes=(recv_counts * es).tolist(),
                                        input_split_sizes=(send_counts * es).tolist(), group=self.group)

    def _all_reduce_dev(self, t):
        """In-place sum over the ranks of a device tensor, on the control group (the data group's stream may be
        full of queued exchange batches); host-staged under gloo."""
        if self.host_staged and t.is_cuda:
            h = t.cpu()
            self.dist.all_reduce(h, group=self.ctl)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, group=self.ctl)
        return t

    def _search_cuts(self, bb, nbb: int, d: RadixDigits, beg, end, pre_lo, pre_hi, rank_in, top_digit: int):
        """Exact cut of sorted runs at global ranks.  Boundary i: this rank's run [beg[i], end[i]) of `bb` (sorted by
        mapped key); the union of the ranks' runs is cut below its rank_in[i]-th element in (key, rank, position) order;
        the digits from `top_digit` up are already fixed in (pre_lo[i], pre_hi[i]).  Returns (less, leq): this rank's
        counts of keys below / not above the boundary key, [nb] each.  One 256-way step per digit (mod.rs:110-120's
        cursors, found by search instead of by scan); with the HIP backend the steps run on the device, stream-ordered
        (count kernel -> all-reduce -> pick kernel), and the host reads ONE result at the end."""
        be = self.backend
        nb = len(rank_in)
        if hasattr(be, "search_cuts_device"):
            return be.search_cuts_device(bb, nbb, d, beg, end, pre_lo, pre_hi, rank_in, top_digit, self._all_reduce_dev)
        pre_lo, pre_hi = pre_lo.copy(), pre_hi.copy()
        j256 = np.arange(256, dtype=np.uint64)
        for digit in range(top_digit - 1, -1, -1):
            lo = np.repeat(pre_lo[:, None], 256, axis=1)
            hi = np.repeat(pre_hi[:, None], 256, axis=1)
            if digit < 8:
                lo |= j256[None, :] << np.uint64(8 * digit)
            else:
                hi |= j256[None, :] << np.uint64(8 * (digit - 8))
            less, _ = be.bounds_ranges(bb, nbb, d, lo.reshape(-1), hi.reshape(-1), np.repeat(beg, 256), np.repeat(end, 256))
            gl = self._all_reduce_sum(less, bb.device, ctl=True).reshape(nb, 256)
            j = ((gl <= rank_in[:, None]).sum(axis=1) - 1).astype(np.uint64)
            if digit < 8:
                pre_lo |= j << np.uint64(8 * digit)
            else:
                pre_hi |= j << np.uint64(8 * (digit - 8))
        return be.bounds_ranges(bb, nbb, d, pre_lo, pre_hi, beg, end)

    def sort_exchange_first(self, x, d: RadixDigits, n_per_rank: Optional[List[int]] = None, chunks: int = 1,
                            sub_ranges: int = 1, donate: bool = False):
        """Same contract as `sort`: partition by the top digit, exchange once, sort once.  Returns the tensor that
        holds the sorted slice: `x` itself, unless `donate` (below).

        chunks (C) pipelines the exchange with the local sort at the RECEIVER: what a rank receives is cut into C
        consecutive ranges of top-digit values of about equal size; a range is final once it is sorted (the ranges
        are ordered by key), so it is sorted while the ranges behind it are still on the links.
        sub_ranges (K) pipelines the partition pass with the exchange at the SENDER: the slice is partitioned in K
        position sub-ranges (rsx_partition_count_device / rsx_partition_scatter_device); the batches go out range-major
        and sub-range-minor, so the first range's pieces of sub-range 0 are on the links while sub-range 1 is still
        being scattered.  The buckets a slice boundary cuts through travel last: every rank merges its K pieces of
        such a bucket, sorts them, and the exact cut is found by a digit-wise search on the device.
        With K > 1 the receive buffer cannot be `x` (still being read by the scatters): the result lands in a second
        buffer, which is copied back into `x` -- or, with `donate`, handed to the caller in exchange for `x`."""
        import torch
        dist, be = self.dist, self.backend
        es, G, me = d.elem_bytes, self.world, self.rank
        n_local = x.numel() // es
        if n_per_rank is None:
            n_per_rank = self._gather_counts(n_local, x.device)
        n_per_rank = np.asarray(n_per_rank, dtype=np.int64)
        assert n_per_rank[me] == n_local
        K = max(1, min(int(sub_ranges), 16))
        C = max(1, int(chunks))
        staged = self.host_staged and x.is_cuda
        bounds = np.concatenate(([0], np.cumsum(n_per_rank)))
        top = d.key_bytes - 1
        part = self._buf("part", n_local * es)
        recv = x if K == 1 else self._buf("recv", n_local * es)
        self.last_branch = f"exchange-first: {K} sub-range(s) x {C} range(s), " + ("p2p batches" if G > 1 else "single rank")
        self._mark("start")
        # 1. count phase of the partition pass per sub-range (mod.rs:90-109 with chunk == sub-range of a rank)
        HH = self._all_gather_i64(be.partition_count(x, n_local, d, top, K).reshape(-1), x.device).reshape(G, K, 256)
        self._mark("count + gather")
        H = HH.sum(axis=1)                                                       # [G][256]
        lst = np.concatenate([np.zeros((G, K, 1), np.int64), np.cumsum(HH, axis=2)], axis=2)  # bucket v inside block (g, k)
        sk = np.array([[(int(n_per_rank[g]) * k) // K for k in range(K + 1)] for g in range(G)], dtype=np.int64)
        tot = H.sum(axis=0)
        gstart = np.concatenate(([0], np.cumsum(tot)))                           # [257] buckets in global order
        assert gstart[256] == bounds[G]
        # 2. which buckets a slice boundary cuts through (they are cut exactly and travel last); who owns the others
        inside = []
        for b in range(G - 1):
            T = bounds[b + 1]
            v = int(np.searchsorted(gstart[1:], T, side="right"))
            if v < 256 and gstart[v] < T:
                inside.append((b, v))
        Bset = sorted({v for _, v in inside})
        isB = np.zeros(256, dtype=bool)
        isB[Bset] = True
        owner = np.searchsorted(bounds[1:], gstart[:256], side="right")          # rank whose range holds the bucket's start
        owner = np.minimum(owner, G - 1)
        chunk_of = np.zeros(256, dtype=np.int64)
        for h in range(G):
            vs = [v for v in range(256) if not isB[v] and owner[v] == h and tot[v] > 0]
            whole = int(tot[vs].sum()) if vs else 0
            run = 0
            for v in vs:
                chunk_of[v] = min(C - 1, (run * C) // max(1, whole))
                run += int(tot[v])
        # 3. layout of every rank's slice after the exchange: regions in bucket order -- a boundary bucket's share
        #    ("B", v) or a range of whole buckets ("W", c); every region is sorted by itself once it is complete
        def regions_of(h):
            regs, off = [], 0
            for v in range(256):
                if tot[v] == 0:
                    continue
                if isB[v]:
                    ln = int(min(gstart[v + 1], bounds[h + 1]) - max(gstart[v], bounds[h]))
                    if ln > 0:
                        regs.append([("B", v), off, ln])
                        off += ln
                elif owner[v] == h:
                    key = ("W", int(chunk_of[v]))
                    if regs and regs[-1][0] == key:
                        regs[-1][2] += int(tot[v])
                    else:
                        regs.append([key, off, int(tot[v])])
                    off += int(tot[v])
            assert off == n_per_rank[h], (h, off, n_per_rank[h])
            return {r[0]: (r[1], r[2]) for r in regs}
        my_regs = regions_of(me)

        def piece(buf, off, ln):
            return buf[int(off) * es:int(off + ln) * es]

        # whole buckets: size_w[g][k][h][c] elements of block (g, k) go to rank h in range c, one contiguous piece
        size_w = np.zeros((G, K, G, C), dtype=np.int64)
        first_v = np.full((G, C), -1, dtype=np.int64)
        for v in range(256):
            if isB[v] or tot[v] == 0:
                continue
            h, c = int(owner[v]), int(chunk_of[v])
            size_w[:, :, h, c] += HH[:, :, v]
            if first_v[h, c] < 0:
                first_v[h, c] = v

        def batch_whole(c, k):
            """(ops, landing, local copies) of range c / sub-range k: my pieces out, the other ranks' pieces in."""
            ops, landing = [], []
            for step in range(G):
                h, g = (me + step) % G, (me - step) % G
                ln = int(size_w[me, k, h, c])
                if ln:
                    src = piece(part, sk[me, k] + lst[me, k, first_v[h, c]], ln)
                    if h == me:
                        dst_off = my_regs[("W", c)][0] + int(size_w[:me, :, me, c].sum()) + int(size_w[me, :k, me, c].sum())
                        piece(recv, dst_off, ln).copy_(src)
                    else:
                        ops.append(dist.P2POp(dist.isend, src.cpu() if staged else src, h, group=self.group))
                ln = int(size_w[g, k, me, c])
                if ln and g != me:
                    dst_off = my_regs[("W", c)][0] + int(size_w[:g, :, me, c].sum()) + int(size_w[g, :k, me, c].sum())
                    dst = piece(recv, dst_off, ln)
                    if staged:
                        host = torch.empty(dst.numel(), dtype=torch.uint8)
                        landing.append((dst, host))
                        dst = host
                    ops.append(dist.P2POp(dist.irecv, dst, g, group=self.group))
            return (dist.batch_isend_irecv(ops) if ops else []), landing

        # 4. scatter phase per sub-range (mod.rs:110-168); range 0 of a sub-range goes out as soon as it is scattered
        works = {}
        for k in range(K):
            if n_local:
                be.partition_scatter(x, part, n_local, d, top, K, k)
            works[(0, k)] = batch_whole(0, k)
        for c in range(1, C):
            for k in range(K):
                works[(c, k)] = batch_whole(c, k)
        self._mark("scatter + batches issued")
        # 5. boundary buckets: merge my K pieces of each, sort, cut exactly, exchange the shares
        bworks = None
        if inside:
            myB = [v for v in Bset if H[me, v] > 0]
            if myB:
                bb = torch.cat([piece(part, sk[me, k] + lst[me, k, v], HH[me, k, v]) for v in myB for k in range(K)])
            else:
                bb = part[:0]
            nbb = bb.numel() // es
            if nbb > 1:  # one sort: the buckets differ in their top digit, so they stay apart and in order
                be.sort(bb, self._buf("bb_scratch", nbb * es), nbb, d)
            boff = {}
            o = 0
            for v in Bset:
                boff[v] = o
                o += int(H[me, v])
            nb = len(inside)
            rank_in = np.array([bounds[b + 1] - gstart[v] for b, v in inside], dtype=np.int64)
            pre_lo = np.zeros(nb, dtype=np.uint64)
            pre_hi = np.zeros(nb, dtype=np.uint64)
            for i, (_, v) in enumerate(inside):
                if top < 8:
                    pre_lo[i] = np.uint64(v) << np.uint64(8 * top)
                else:
                    pre_hi[i] = np.uint64(v) << np.uint64(8 * (top - 8))
            beg = np.array([boff[v] for _, v in inside], dtype=np.uint64)
            end = np.array([boff[v] + int(H[me, v]) for _, v in inside], dtype=np.uint64)
            l_me, q_me = self._search_cuts(bb, nbb, d, beg, end, pre_lo, pre_hi, rank_in, top)
            M = self._all_gather_i64(np.stack([l_me, q_me]).astype(np.int64).reshape(-1), x.device, ctl=True).reshape(G, 2, nb)
            less, eq = M[:, 0, :], M[:, 1, :] - M[:, 0, :]
            need = rank_in - less.sum(axis=0)  # elements equal to the boundary key that go below the cut
            before = np.cumsum(eq, axis=0) - eq
            cut = less + np.clip(need[None, :] - before, 0, eq)  # [G][nb]; ties: lower rank first (stability)
            self._mark("boundary buckets: sort + exact cuts")
            # share[g][v][h] = (offset in g's sorted bucket v, length) going to rank h
            ops, landing = [], []
            for v in Bset:
                bs = [i for i, (_, vv) in enumerate(inside) if vv == v]  # boundaries inside v, ascending
                b0 = inside[bs[0]][0]
                edges = np.concatenate([np.zeros((G, 1), np.int64), cut[:, bs], H[:, v:v + 1]], axis=1)  # [G][m + 2]
                for j in range(len(bs) + 1):
                    h = b0 + j  # the ranks b0 .. b0 + m share bucket v
                    ln_g = edges[:, j + 1] - edges[:, j]
                    assert (ln_g >= 0).all()
                    if h == me:
                        if ("B", v) not in my_regs:
                            assert ln_g.sum() == 0
                            continue
                        roff, rlen = my_regs[("B", v)]
                        assert ln_g.sum() == rlen, (v, ln_g.sum(), rlen)
                        for g in range(G):
                            if not ln_g[g]:
                                continue
                            dst = piece(recv, roff + int(ln_g[:g].sum()), ln_g[g])
                            if g == me:
                                dst.copy_(piece(bb, boff[v] + edges[me, j], ln_g[me]))
                            else:
                                if staged:
                                    host = torch.empty(dst.numel(), dtype=torch.uint8)
                                    landing.append((dst, host))
                                    dst = host
                                ops.append(dist.P2POp(dist.irecv, dst, g, group=self.group))
                    elif ln_g[me]:
                        src = piece(bb, boff[v] + edges[me, j], ln_g[me])
                        ops.append(dist.P2POp(dist.isend, src.cpu() if staged else src, h, group=self.group))
            # one batch: per peer the sends and the receives pair up in bucket order on both sides
            bworks = ((dist.batch_isend_irecv(ops) if ops else []), landing)
        # 6. every region is sorted as soon as it is complete (stable LSD sort of mod.rs:84-169 by the whole key)
        scratch_len = max([ln for _, ln in my_regs.values()], default=0)
        scratch = self._buf("sort_scratch", scratch_len * es)

        def finish(reqs_landing, key):
            for reqs, landing in reqs_landing:
                for r in reqs:
                    r.wait()  # (RCCL: the compute stream waits, not the host)
                for dst, host in landing:
                    dst.copy_(host)
            if key in my_regs:
                off, ln = my_regs[key]
                if ln > 1:
                    be.sort(piece(recv, off, ln), scratch[:ln * es], ln, d)
        for c in range(C):
            finish([works[(c, k)] for k in range(K)], ("W", c))
        if bworks is not None:
            finish([bworks], None)
            for v in Bset:
                finish([], ("B", v))
        be.finish()
        self._mark("exchange + sorts")
        self._report()
        if recv is not x:
            if donate:
                self._bufs["recv"] = x  # the caller's buffer is our scratch from now on
                return recv
            x.copy_(recv)
        return x

    # ---- verification of a sharded result (bench.py's N > 1 check and the tests share it) ----------------
    def checksum(self, x, d: RadixDigits) -> int:
        """Multiset checksum of this rank's slice (rsx_verify_device out[1]), taken BEFORE a sort."""
        import torch
        ctx = self.backend.ctx
        n = x.numel() // d.elem_bytes
        out = torch.zeros(3, dtype=torch.int64, device=x.device)
        ctx.verify_device(x.data_ptr(), n, d, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        return int(out[1].item())

    def verify(self, x, d: RadixDigits, checksum_before: int) -> None:
        """Collective.  Raises AssertionError unless (a) every rank's slice is in order and stable (payload =
        global index), (b) the multiset is unchanged (checksums summed over the ranks mod 2^64), (c) the last key of
        rank r does not exceed the first key of rank r + 1 (empty slices are stepped over)."""
        import torch
        dist, ctx = self.dist, self.backend.ctx
        es = d.elem_bytes
        n = x.numel() // es
        cdev = torch.device("cpu") if self.host_staged else x.device
        out = torch.zeros(3, dtype=torch.int64, device=x.device)
        ctx.verify_device(x.data_ptr(), n, d, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        v = out.cpu().tolist()
        assert v[0] == 0 and v[2] == 0, f"rank {self.rank}: slice not sorted / not stable: {v}"
        before = checksum_before - (1 << 64) if checksum_before >= (1 << 63) else checksum_before
        sums = torch.tensor([before, v[1]], dtype=torch.int64).to(cdev)
        dist.all_reduce(sums, group=self.group)  # int64 wrap-around == the checksum's arithmetic mod 2^64
        assert sums[0].item() == sums[1].item(), "multiset checksum changed across the exchange"

        def mapped(e):  # order-preserving integer of one element's key (radix_digits.rs via RadixDigits.get_digit)
            return sum(d.get_digit(bytes(e.tolist()), i) << (8 * i) for i in range(d.key_bytes))

        def limbs(k):  # 128-bit key as four 32-bit limbs, most significant first (int64 tensors carry them)
            return [(k >> s) & 0xFFFFFFFF for s in (96, 64, 32, 0)]
        if n:
            edge = torch.cat([x[:es], x[(n - 1) * es:n * es]]).cpu()
            mine = torch.tensor([1] + limbs(mapped(edge[:es])) + limbs(mapped(edge[es:])), dtype=torch.int64)
        else:
            mine = torch.zeros(9, dtype=torch.int64)
        allk = [torch.zeros(9, dtype=torch.int64, device=cdev) for _ in range(self.world)]
        dist.all_gather(allk, mine.to(cdev), group=self.group)
        last = None
        for r, gathered in enumerate(allk):
            a = [int(t) for t in gathered.cpu().tolist()]
            if not a[0]:
                continue
            first_k = a[1] << 96 | a[2] << 64 | a[3] << 32 | a[4]
            last_k = a[5] << 96 | a[6] << 64 | a[7] << 32 | a[8]
            assert last is None or last[1] <= first_k, f"order broken between rank {last[0]} and {r}"
            last = (r, last_k)

    def sort(self, x, d: RadixDigits, n_per_rank: Optional[List[int]] = None):
        """x: this rank's slice as a contiguous uint8 tensor (n_local * elem_bytes).  In place."""
        dist, be = self.dist, self.backend
        es = d.elem_bytes
        n_local = x.numel() // es
        if n_per_rank is None:
            n_per_rank = self._gather_counts(n_local, x.device)
        n_per_rank = np.asarray(n_per_rank, dtype=np.int64)
        assert n_per_rank[self.rank] == n_local
        part = self._buf("part", n_local * es)
        recv = self._buf("recv", n_local * es)
        hist = be.zeros_u64(256)
        import torch
        gathered = [torch.zeros(256, dtype=torch.int64, device=x.device) for _ in range(self.world)]
        for digit in range(d.key_bytes):  # mod.rs:84
            be.partition(x, part, n_local, d, digit, hist)  # count + local stable scatter
            if self.host_staged and hist.is_cuda:
                hcpu = hist.cpu()
                gl = [torch.zeros(256, dtype=torch.int64) for _ in range(self.world)]
                dist.all_gather(gl, hcpu, group=self.group)
                H = torch.stack(gl).numpy()
            else:
                dist.all_gather(gathered, hist, group=self.group)  # the G x 256 counts
                H = torch.stack(gathered).cpu().numpy()
            send_counts, recv_counts, segs = exchange_plan(H, n_per_rank, self.rank)
            assert send_counts.sum() == n_local and recv_counts.sum() == n_local
            if self.host_staged and part.is_cuda:
                rc = torch.empty(n_local * es, dtype=torch.uint8)
                dist.all_to_all_single(rc, part.cpu(), output_split_sizes=(recv_counts * es).tolist(),
                                       input_split_sizes=(send_counts * es).tolist(), group=self.group)
                recv.copy_(rc)
            else:
                dist.all_to_all_single(recv, part, output_split_sizes=(recv_counts * es).tolist(),
                                       input_split_sizes=(send_counts * es).tolist(), group=self.group)
            nseg = segs.shape[0]
            if nseg:
                so = be.to_device_i64(np.ascontiguousarray(segs[:, 0]))
                do = be.to_device_i64(np.ascontiguousarray(segs[:, 1]))
                ln = be.to_device_i64(np.ascontiguousarray(segs[:, 2]))
                be.segmented_copy(recv, x, es, so, do, ln, nseg)
        be.finish()
