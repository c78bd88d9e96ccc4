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
    """`radix_sort` over a slice-per-rank array.  All ranks call `sort` collectively."""

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

    def _buf(self, name: str, nbytes: int):
        b = self._bufs.get(name)
        if b is None or b.numel() < nbytes:
            b = self._bufs[name] = self.backend.empty_bytes(nbytes)
        return b[:nbytes]

    def sort(self, x, d: RadixDigits, n_per_rank: Optional[List[int]] = None):
        """x: this rank's slice as a contiguous uint8 tensor (n_local * elem_bytes).  In place."""
        dist, be = self.dist, self.backend
        es = d.elem_bytes
        n_local = x.numel() // es
        if n_per_rank is None:
            import torch
            t = torch.tensor([n_local], dtype=torch.int64, device=x.device)
            allt = [torch.zeros_like(t) for _ in range(self.world)]
            dist.all_gather(allt, t, group=self.group)
            n_per_rank = [int(a.item()) for a in allt]
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
