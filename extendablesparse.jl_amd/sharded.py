"""Column-range sharded assembly across the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).  Every rank
appends whatever updates its part of the assembly loop produces -- like one reference buffer per
`tid` (src/matrix/genericmtextendablesparsematrixcsc.jl:87-99) -- and flush! routes every pending
entry to the rank that owns its column with ONE all-to-all-v before the local flush:

    owner(col) = floor((col-1) * P / n)                      contiguous column ranges
    1. stable partition of the pending entries by owner       (esp_shard_export, HIP)
    2. all_to_all_single of the per-destination counts        (P x 8 B)
    3. all_to_all_single of the 8-byte keys and 8-byte values (RCCL; per link: bytes_to_peer/153 GB/s)
    4. local flush of the received entries                    (same kernels as the 1-GPU path)
    5. all_gather of the local nnz -> colptr offsets of the global CSC

Received chunks are ordered by source rank and keep the source's append order, so the ordered
fold stays deterministic: the result equals ONE buffer fed the streams of rank 0, 1, ... in turn.

The exchange logic is independent of where the entries live: `backend` supplies the local
operations.  HipShardBackend is the product (device memory, C ABI); tests drive the same class
with a CPU backend over gloo.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from ._lib import ESP_FLUSH_ROUTED
from .matrix import ExtendableSparseMatrix, SparseMatrixCSC


def owner_ranges(n, P):
    """0-based column range [c0,c1) of every shard: owner(col0) = floor(col0*P/n)."""
    bounds = [-(-r * n // P) for r in range(P + 1)]  # ceil(r*n/P)
    return [(bounds[r], bounds[r + 1]) for r in range(P)]


# torch 2.10+rocm7.0 / RCCL 2.26: one all_to_all_single call that moves more than 2^27 8-byte
# elements delivers only part of the data (measured on MI355X, tools: tests/test_gpu_parity.py
# ::test_all_to_all_large_message).  The exchange is therefore issued in rounds of bounded size.
A2A_MAX_ELEMS = 1 << 26


def all_to_all_v(dist, out, inp, out_splits, in_splits, group=None, max_elems=A2A_MAX_ELEMS):
    """all-to-all-v of 8-byte elements in rounds of at most max_elems per call; `out` keeps the
    (source rank, source order) layout of a single all_to_all_single."""
    import torch
    P = len(in_splits)
    big = max(max(in_splits, default=0), max(out_splits, default=0))
    t = torch.tensor([big], dtype=torch.int64, device=inp.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    big = int(t.item())
    C = max(1, max_elems // P)
    if big <= C:
        dist.all_to_all_single(out, inp, out_splits, in_splits, group=group)
        return 1
    in_off = np.concatenate([[0], np.cumsum(in_splits)]).astype(np.int64)
    out_off = np.concatenate([[0], np.cumsum(out_splits)]).astype(np.int64)
    rounds = -(-big // C)
    for q in range(rounds):
        in_q = [int(min(max(c - q * C, 0), C)) for c in in_splits]
        out_q = [int(min(max(c - q * C, 0), C)) for c in out_splits]
        send = torch.cat([inp[in_off[d] + q * C: in_off[d] + q * C + in_q[d]] for d in range(P)])
        recv = torch.empty(sum(out_q), dtype=out.dtype, device=out.device)
        dist.all_to_all_single(recv, send, out_q, in_q, group=group)
        pos = 0
        for r in range(P):
            out[out_off[r] + q * C: out_off[r] + q * C + out_q[r]] = recv[pos: pos + out_q[r]]
            pos += out_q[r]
    return rounds


class _DevArray:
    """__cuda_array_interface__ view of library-owned device memory (no copy, no ownership)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def _wrap_device(torch, ptr, n, dtype, device):
    if n == 0:
        return torch.empty(0, dtype=dtype, device=device)
    return torch.as_tensor(_DevArray(ptr, n, "<i8" if dtype == torch.int64 else "<f8"), device=device)


class HipShardBackend:
    """Local operations of one shard on its GPU, through the C ABI."""

    def __init__(self, m, n, device=0, capacity_hint=0):
        import torch
        self.torch = torch
        self.device = torch.device("cuda", device)
        self.A = ExtendableSparseMatrix(m, n, device=device, capacity_hint=capacity_hint)
        self.m, self.n = int(m), int(n)

    @property
    def matrix(self):
        return self.A

    def pending(self):
        return self.A.nnznew()

    def empty(self, count, dtype):
        return self.torch.empty(int(count), dtype=dtype, device=self.device)

    def shard_counts(self, P):
        d = self.A._d
        d.commit()
        counts = np.zeros(P, np.int64)
        d.ck(d.lib.esp_shard_counts(d.h, P, counts.ctypes.data_as(C.c_void_p)))
        return counts

    def shard_export(self, P):
        torch = self.torch
        d = self.A._d
        d.commit()
        E = d.pending()
        keys = torch.empty(E, dtype=torch.int64, device=self.device)
        vals = torch.empty(E, dtype=torch.float64, device=self.device)
        offsets = np.zeros(P + 1, np.int64)
        d.ck(d.lib.esp_shard_export(d.h, P, C.c_void_p(keys.data_ptr()), C.c_void_p(vals.data_ptr()),
                                    offsets.ctypes.data_as(C.c_void_p)))
        return keys, vals, offsets

    def replace_pending(self, pieces):
        """pending := concatenation of the (keys, vals) device pieces, in order."""
        d = self.A._d
        self.torch.cuda.synchronize(self.device)  # the collective ran on torch's stream
        d.ck(d.lib.esp_clear_pending(d.h))
        for keys, vals in pieces:
            if keys.numel():
                d.ck(d.lib.esp_append_packed(d.h, C.c_void_p(keys.data_ptr()), C.c_void_p(vals.data_ptr()), keys.numel()))
        d.ck(d.lib.esp_synchronize(d.h))  # the copies are done before the tensors may be freed
        self.A._touch()

    def set_column_window(self, col_lo, col_hi):
        self.A.set_column_window(col_lo, col_hi)

    # -- in-place exchange (the own chunk is not copied): see esp_shard_exchange_begin
    def exchange_begin(self, P, me, recv_lower, recv_higher):
        torch = self.torch
        d = self.A._d
        d.commit()
        pk, pv = C.c_void_p(), C.c_void_p()
        soff = np.zeros(P + 1, np.int64)
        d.ck(d.lib.esp_shard_exchange_begin(d.h, P, me, int(recv_lower), int(recv_higher), C.byref(pk), C.byref(pv),
                                            soff.ctypes.data_as(C.c_void_p)))
        n = int(soff[-1])
        keys = _wrap_device(torch, pk.value, n, torch.int64, self.device)
        vals = _wrap_device(torch, pv.value, n, torch.float64, self.device)
        self.A._touch()
        return keys, vals, soff

    def exchange_place(self, position, keys, vals):
        d = self.A._d
        if keys.numel():
            self.torch.cuda.synchronize(self.device)
            d.ck(d.lib.esp_shard_exchange_place(d.h, int(position), C.c_void_p(keys.data_ptr()), C.c_void_p(vals.data_ptr()),
                                                keys.numel()))

    def flush(self):
        self.A.flush()
        return self.A._d.nnz()

    def local_csc(self):
        return self.A.sparse()


class ShardedExtendableSparseMatrix:
    """ExtendableSparseMatrix whose columns are sharded over the ranks of a process group."""

    def __init__(self, m, n, backend, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.P = dist.get_world_size(group)
        self.m, self.n = int(m), int(n)
        self.backend = backend
        self.ranges = owner_ranges(self.n, self.P)
        self.local_nnz = 0
        self.nnz_offsets = np.zeros(self.P + 1, np.int64)
        c0, c1 = self.ranges[self.rank]
        if hasattr(backend, "set_column_window") and c1 > c0:
            backend.set_column_window(c0 + 1, c1)  # after the exchange every pending column is owned

    # -- updates go to the local buffer, whatever their column (like xmatrices[tid])
    @property
    def local(self):
        return self.backend.matrix

    def updateindex(self, op, v, i, j):
        self.local.updateindex(op, v, i, j)

    def rawupdateindex(self, op, v, i, j, tid=1):
        self.local.rawupdateindex(op, v, i, j)

    def __setitem__(self, ij, v):
        self.local[ij] = v

    def append(self, kind, I, J, V, op="+", kinds=None):
        self.local.append(kind, I, J, V, op, kinds)

    # -- the exchange + local flush
    def flush(self):
        import torch
        dist, P, be = self.dist, self.P, self.backend
        me = self.rank
        if hasattr(be, "exchange_begin"):
            # counts first, then the in-place partition (own chunk stays on the device where it is)
            counts = be.shard_counts(P)
            send_counts = torch.from_numpy(counts.astype(np.int64))
            dev = be.device
            rc = torch.empty(P, dtype=torch.int64, device=dev)
            dist.all_to_all_single(rc, send_counts.to(dev), group=self.group)
            out_splits = rc.cpu().tolist()
            in_splits = counts.tolist()
            out_x = list(out_splits)
            out_x[me] = 0
            in_x = list(in_splits)
            in_x[me] = 0
            lower, higher = int(sum(out_x[:me])), int(sum(out_x[me + 1:]))
            skeys, svals, soff = be.exchange_begin(P, me, lower, higher)
            rkeys = be.empty(lower + higher, torch.int64)
            rvals = be.empty(lower + higher, torch.float64)
            all_to_all_v(dist, rkeys, skeys, out_x, in_x, self.group)
            all_to_all_v(dist, rvals, svals, out_x, in_x, self.group)
            be.exchange_place(0, rkeys[:lower], rvals[:lower])
            be.exchange_place(lower + in_splits[me], rkeys[lower:], rvals[lower:])
            self.exchanged = (int(sum(in_splits)), int(sum(out_splits)))
            self.sent_off_rank = int(sum(in_x))
        else:
            self._flush_exchange_generic()
        self.local_nnz = be.flush()
        # global colptr offsets: exclusive scan of the per-shard nnz
        dev = getattr(be, "device", None)
        mine = torch.tensor([self.local_nnz], dtype=torch.int64, device=dev if dev is not None else "cpu")
        allnnz = [torch.empty_like(mine) for _ in range(P)]
        dist.all_gather(allnnz, mine, group=self.group)
        counts = np.array([int(t.item()) for t in allnnz], np.int64)
        self.nnz_offsets = np.concatenate([[0], np.cumsum(counts)])
        return self


    def _flush_exchange_generic(self):
        """Exchange through export buffers (any backend): used by the CPU tests."""
        import torch
        dist, P, be = self.dist, self.P, self.backend
        keys, vals, offsets = be.shard_export(P)
        send_counts = torch.from_numpy(np.diff(offsets).astype(np.int64))
        recv_counts = torch.empty(P, dtype=torch.int64)
        dev = keys.device
        if dev.type == "cuda":  # NCCL/RCCL moves device tensors only
            sc, rc = send_counts.to(dev), recv_counts.to(dev)
            dist.all_to_all_single(rc, sc, group=self.group)
            recv_counts = rc.cpu()
        else:
            dist.all_to_all_single(recv_counts, send_counts, group=self.group)
        in_splits = send_counts.tolist()
        out_splits = recv_counts.tolist()
        # the chunk a rank owns itself never enters the collective (in a slab-wise assembly that is
        # almost everything): it is appended straight from the export buffer, in rank position
        me = self.rank
        own = in_splits[me]
        assert out_splits[me] == own
        in_x = list(in_splits)
        out_x = list(out_splits)
        in_x[me] = 0
        out_x[me] = 0
        own_lo = int(offsets[me])
        if own:
            skeys = torch.cat([keys[:own_lo], keys[own_lo + own:]]) if sum(in_x) else keys[:0]
            svals = torch.cat([vals[:own_lo], vals[own_lo + own:]]) if sum(in_x) else vals[:0]
        else:
            skeys, svals = keys, vals
        nrecv = int(sum(out_x))
        rkeys = be.empty(nrecv, torch.int64)
        rvals = be.empty(nrecv, torch.float64)
        all_to_all_v(dist, rkeys, skeys, out_x, in_x, self.group)
        all_to_all_v(dist, rvals, svals, out_x, in_x, self.group)
        self.exchanged = (int(sum(in_splits)), int(sum(out_splits)))
        self.sent_off_rank = int(sum(in_x))
        lower = int(sum(out_x[:me]))
        be.replace_pending([(rkeys[:lower], rvals[:lower]),
                            (keys[own_lo:own_lo + own], vals[own_lo:own_lo + own]),
                            (rkeys[lower:], rvals[lower:])])

    def nnz(self):
        return int(self.nnz_offsets[-1])

    def local_slice(self):
        """This shard's part of the global CSC: (c0, c1, colptr[c0..c1] global 1-based, rowval, nzval)."""
        csc = self.backend.local_csc()
        c0, c1 = self.ranges[self.rank]
        colptr = csc.colptr[c0:c1 + 1] + self.nnz_offsets[self.rank]
        if c0 > 0:
            assert csc.colptr[c0] == 1, "entries left of the owned column range"
        assert csc.colptr[c1] == csc.colptr[-1], "entries right of the owned column range"
        return c0, c1, colptr, csc.rowval, csc.nzval

    def gather_sparse(self, dst=0):
        """Host-visible global SparseMatrixCSC on rank `dst` (None elsewhere); for checks/small sizes."""
        piece = self.local_slice()
        out = [None] * self.P if self.rank == dst else None
        self.dist.gather_object(piece, out, dst=dst, group=self.group)
        if self.rank != dst:
            return None
        colptr = np.ones(self.n + 1, np.int64)
        rows, vals = [], []
        for (c0, c1, cp, rv, nz) in out:
            colptr[c0:c1 + 1] = cp
            rows.append(rv)
            vals.append(nz)
        colptr[-1] = self.nnz_offsets[-1] + 1
        return SparseMatrixCSC(self.m, self.n, colptr, np.concatenate(rows), np.concatenate(vals))
