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

    def _all_reduce_sum(self, a: np.ndarray, device) -> np.ndarray:
        import torch
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64))
        if not self.host_staged:
            t = t.to(device)
        self.dist.all_reduce(t, group=self.group)
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

    def _all_gather_i64(self, a: np.ndarray, device) -> np.ndarray:
        """[G, len(a)] int64: every rank's vector."""
        import torch
        src = torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64))
        if self.host_staged:
            out = [torch.zeros_like(src) for _ in range(self.world)]
            self.dist.all_gather(out, src, group=self.group)
            return torch.stack(out).numpy()
        src = src.to(device)
        out = [torch.zeros_like(src) for _ in range(self.world)]
        self.dist.all_gather(out, src, group=self.group)
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

    def sort_exchange_first(self, x, d: RadixDigits, n_per_rank: Optional[List[int]] = None, chunks: int = 1):
        """Same contract as `sort`.  Partition by the top digit, exchange once, sort once.

        chunks > 1 pipelines the exchange with the local sort: what a rank receives is cut into `chunks`
        consecutive ranges of top-digit values of about equal size; the pieces of range c travel as one
        batch of sends/receives, and range c is sorted (it is final: the ranges are ordered by key) while
        the batches behind it are still on the links.  xGMI moves a slice more slowly than the GPU sorts
        it, so this hides the local sort behind the exchange but for the last range."""
        be = self.backend
        es, G, me = d.elem_bytes, self.world, self.rank
        n_local = x.numel() // es
        if n_per_rank is None:
            n_per_rank = self._gather_counts(n_local, x.device)
        n_per_rank = np.asarray(n_per_rank, dtype=np.int64)
        assert n_per_rank[me] == n_local
        bounds = np.concatenate(([0], np.cumsum(n_per_rank)))
        part = self._buf("part", n_local * es)
        top = d.key_bytes - 1
        self._mark("start")
        # 1. one stable partition pass by the most significant digit (count + scatter, mod.rs:90-168)
        hist = be.zeros_u64(256)
        if n_local:
            be.partition(x, part, n_local, d, top, hist)
        self._mark("partition")
        H = self._all_gather_i64(hist.cpu().numpy() if hasattr(hist, "cpu") else np.asarray(hist), x.device)  # [G][256]
        self._mark("gather-counts")
        # 2. the buckets in global order (digit-major, rank-minor: mod.rs:110-120 with chunk == rank)
        lstart = np.concatenate([np.zeros((G, 1), np.int64), np.cumsum(H, axis=1)], axis=1)  # [G][257]
        tot = H.sum(axis=0)
        gstart = np.concatenate(([0], np.cumsum(tot)))  # [257]
        split = np.zeros((G, G + 1), dtype=np.int64)
        split[:, G] = n_per_rank
        inside = []  # (boundary, bucket) for boundaries strictly inside a bucket
        for b in range(G - 1):
            T = bounds[b + 1]
            v = int(np.searchsorted(gstart[1:], T, side="right"))  # first bucket that ends above T
            if v == 256:
                split[:, b + 1] = n_per_rank
            elif gstart[v] == T:
                split[:, b + 1] = lstart[:, v]
            else:
                inside.append((b, v))
        # boundary buckets: every rank sorts its piece (tmp piece in place, the old slice as scratch)
        def piece(buf, v):
            return buf[lstart[me, v] * es: lstart[me, v + 1] * es]
        # (ONE sort over the concatenation of the pieces rather than one small sort per bucket: a piece holds one
        # top-digit value, so sorting them together by the whole key keeps every piece where it is and sorts
        # it -- and a small sort is mostly launch overhead: ~0.25 ms each, G-1 of them)
        vs = [v for v in sorted({v for _, v in inside}) if lstart[me, v + 1] - lstart[me, v] > 0]
        if len(vs) == 1:
            ln = int(lstart[me, vs[0] + 1] - lstart[me, vs[0]])
            if ln > 1:
                be.sort(piece(part, vs[0]), piece(x, vs[0]), ln, d)
        elif vs:
            import torch
            both = torch.cat([piece(part, v) for v in vs])
            total = both.numel() // es
            be.sort(both, self._buf("sort_scratch", total * es), total, d)
            off = 0
            for v in vs:
                ln = int(lstart[me, v + 1] - lstart[me, v]) * es
                piece(part, v).copy_(both[off:off + ln])
                off += ln
        self._mark("sort-boundary-buckets")
        if inside:
            nb = len(inside)
            rank_in = np.array([bounds[b + 1] - gstart[v] for b, v in inside], dtype=np.int64)
            pre_lo = np.zeros(nb, dtype=np.uint64)
            pre_hi = np.zeros(nb, dtype=np.uint64)
            for i, (_, v) in enumerate(inside):
                if top < 8:
                    pre_lo[i] = np.uint64(v) << np.uint64(8 * top)
                else:
                    pre_hi[i] = np.uint64(v) << np.uint64(8 * (top - 8))
            j256 = np.arange(256, dtype=np.uint64)
            # every boundary's queries are answered inside its own bucket's sorted piece: one call per digit
            beg = np.array([lstart[me, v] for _, v in inside], dtype=np.uint64)
            end = np.array([lstart[me, v + 1] for _, v in inside], dtype=np.uint64)
            for digit in range(top - 1, -1, -1):
                lo = np.repeat(pre_lo[:, None], 256, axis=1)
                hi = np.repeat(pre_hi[:, None], 256, axis=1)
                if digit < 8:
                    lo |= j256[None, :] << np.uint64(8 * digit)
                else:
                    hi |= j256[None, :] << np.uint64(8 * (digit - 8))
                less, _ = be.bounds_ranges(part, n_local, d, lo.reshape(-1), hi.reshape(-1), np.repeat(beg, 256), np.repeat(end, 256))
                gl = self._all_reduce_sum(less, x.device).reshape(nb, 256)
                j = ((gl <= rank_in[:, None]).sum(axis=1) - 1).astype(np.uint64)
                if digit < 8:
                    pre_lo |= j << np.uint64(8 * digit)
                else:
                    pre_hi |= j << np.uint64(8 * (digit - 8))
            l_me, q_me = be.bounds_ranges(part, n_local, d, pre_lo, pre_hi, beg, end)
            mine = np.stack([l_me, q_me]).astype(np.int64)
            M = self._all_gather_i64(mine.reshape(-1), x.device).reshape(G, 2, nb)
            less, eq = M[:, 0, :], M[:, 1, :] - M[:, 0, :]
            need = rank_in - less.sum(axis=0)  # elements equal to the boundary key that go below the cut
            before = np.cumsum(eq, axis=0) - eq
            take = np.clip(need[None, :] - before, 0, eq)  # ties: lower rank first (stability)
            for i, (b, v) in enumerate(inside):
                split[:, b + 1] = lstart[:, v] + less[:, i] + take[:, i]
        send_counts = np.diff(split[me])
        recv_counts = split[:, me + 1] - split[:, me]
        assert send_counts.sum() == n_local and recv_counts.sum() == n_local, (send_counts, recv_counts)
        self._mark("exact-cuts")
        if chunks > 1 and G > 1:
            self._pipelined_exchange_and_sort(x, part, d, split, lstart, chunks)
            be.finish()
            self._mark("exchange+sort (pipelined)")
            self._report()
            return
        # 3. the one exchange, straight back into the slice; 4. one local sort
        self._exchange(part, x, send_counts, recv_counts, es)
        self._mark("exchange")
        if n_local > 1:
            be.sort(x, part, n_local, d)
        be.finish()
        self._mark("sort")
        self._report()

    def _pipelined_exchange_and_sort(self, x, part, d: RadixDigits, split: np.ndarray, lstart: np.ndarray, chunks: int):
        """Steps 3 and 4 of `sort_exchange_first`, overlapped.  split[g][h]..split[g][h+1] of rank g's
        partitioned slice goes to rank h; lstart[g][v] is where top-digit bucket v starts in it.  Every rank
        derives the same plan from these two tables."""
        import torch
        dist, be = self.dist, self.backend
        es, G, me = d.elem_bytes, self.world, self.rank
        n_local = x.numel() // es
        # cnt[g][h][v]: elements of rank g's bucket v that go to rank h (overlap of the bucket with h's range)
        lo = np.maximum(lstart[:, None, :-1], split[:, :-1, None])
        hi = np.minimum(lstart[:, None, 1:], split[:, 1:, None])
        cnt = np.clip(hi - lo, 0, None)  # [G][G][256]
        # ranges of top-digit values, per destination, of about equal size: boundaries by cumulative count
        tot = cnt.sum(axis=0)  # [h][v]
        cum = np.cumsum(tot, axis=1)
        edges = np.zeros((G, chunks + 1), dtype=np.int64)  # bucket index where chunk c of destination h starts
        for h in range(G):
            n_h = cum[h, -1]
            for c in range(1, chunks):
                edges[h, c] = int(np.searchsorted(cum[h], (n_h * c) // chunks, side="right"))
            edges[h, chunks] = 256
            edges[h] = np.maximum.accumulate(edges[h])
        # size[g][h][c]: what g sends to h in batch c (one contiguous piece: buckets are in order inside g's range for h)
        size = np.zeros((G, G, chunks), dtype=np.int64)
        for h in range(G):
            for c in range(chunks):
                size[:, h, c] = cnt[:, h, edges[h, c]:edges[h, c + 1]].sum(axis=1)
        send_off = split[me, :-1, None] + np.cumsum(size[me], axis=1) - size[me]  # [h][c] offsets in `part`
        csize = size[:, me, :].sum(axis=0)  # my chunk sizes
        coff = np.concatenate(([0], np.cumsum(csize)))
        recv_off = coff[None, :-1] + np.cumsum(size[:, me, :], axis=0) - size[:, me, :]  # [g][c] offsets in x
        assert coff[-1] == n_local
        scratch = self._buf("sort_scratch", int(csize.max()) * es if n_local else 0)

        def piece(buf, off, ln):
            return buf[int(off) * es:int(off + ln) * es]

        staged = self.host_staged and x.is_cuda
        works = []
        for c in range(chunks):
            ops, landing = [], []
            for k in range(1, G):  # ring order: every link starts busy
                h, g = (me + k) % G, (me - k) % G
                if size[me, h, c]:
                    src = piece(part, send_off[h, c], size[me, h, c])
                    ops.append(dist.P2POp(dist.isend, src.cpu() if staged else src, h, group=self.group))
                if size[g, me, c]:
                    dst = piece(x, recv_off[g, c], size[g, me, c])
                    if staged:
                        host = torch.empty(dst.numel(), dtype=torch.uint8)
                        landing.append((dst, host))
                        dst = host
                    ops.append(dist.P2POp(dist.irecv, dst, g, group=self.group))
            reqs = dist.batch_isend_irecv(ops) if ops else []
            if size[me, me, c]:  # my own piece: a local copy
                piece(x, recv_off[me, c], size[me, me, c]).copy_(piece(part, send_off[me, c], size[me, me, c]))
            works.append((reqs, landing))
        for c in range(chunks):
            reqs, landing = works[c]
            for r in reqs:
                r.wait()  # (RCCL: the compute stream waits, not the host)
            for dst, host in landing:
                dst.copy_(host)
            if csize[c] > 1:
                be.sort(piece(x, coff[c], csize[c]), scratch[:int(csize[c]) * es], int(csize[c]), d)

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
