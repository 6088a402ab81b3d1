"""TEST INFRASTRUCTURE: a Python MODEL of the sharded flush over torch.distributed.

The product's sharded flush is ONE C call per rank (esp_group_flush: policy in csrc/group_policy.hpp, RCCL transport in
csrc/group.hpp; Python caller: extendablesparse.jl_amd/sharded.py, GroupShardedMatrix).  This module orchestrates the same
esp_shard_* building blocks from Python instead and is what the CPU tests run over gloo with a host stand-in for the
device operations (tests/test_sharded_gloo.py) and what the ranks-as-threads GPU tests drive through the real device
backend (HipShardBackend).  It used to live inside the package; nothing of the product imports it.


One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).  Every rank
appends whatever updates its part of the assembly loop produces -- like one reference buffer per
`tid` (src/matrix/genericmtextendablesparsematrixcsc.jl:87-99) -- and flush! routes every pending
entry to the rank that owns its column with ONE all-to-all-v before the local flush:

    owner(col) = floor((col-1) * P / n)                      contiguous column ranges

Partitioned exchange (streams an assembly loop emits; `last_exchange == "partitioned"`):
    1. ONE stable pass per rank partitions its pending entries by (owner, digit inside the owner's
       column range): the owner split and the first partition pass of the local flush at once
       (esp_shard_partition, HIP, run-based single pass)
    2. all_gather of (applicable?, entries per owner)        -> every rank takes the same decision
    3. all_to_all of keys, values (ranges of the other owners; the own range is not touched) and of
       the per-digit counts                                   (RCCL; per link: bytes_to_peer/153 GB/s);
       a small exchange (slab-wise assembly) travels as ONE message per destination
       [counts | keys | values]: one collective instead of three
    4. esp_shard_assemble: piece tables -- the bucket kernel reads a segment as the concatenation of
       one piece per source rank, straight from the receive buffers (nothing is copied or re-sorted)
    5. bucket kernel + colptr (unchanged), all_gather of the local nnz -> global colptr offsets
Plain exchange (any stream, and whenever some rank reports "not applicable"): stable partition by owner
(esp_shard_exchange_begin), all-to-all-v, esp_shard_exchange_place, ordinary local flush.

Either way the received entries are ordered by source rank and keep the source's append order, so the
ordered fold stays deterministic: the result equals ONE buffer fed the streams of rank 0, 1, ... in turn.

The exchange logic is independent of where the entries live: `backend` supplies the local
operations.  HipShardBackend is the product (device memory, C ABI); tests drive the same class
with a CPU backend over gloo.
"""
import ctypes as C

import numpy as np

from esparse_loader import load as _load

_esp = _load()
L = _esp._lib
ESP_FLUSH_ROUTED = L.ESP_FLUSH_ROUTED
ExtendableSparseMatrix = _esp.ExtendableSparseMatrix
SparseMatrixCSC = _esp.SparseMatrixCSC
owner_ranges = _esp.owner_ranges


# torch 2.10+rocm7.0 / RCCL 2.26: one all_to_all_single call that moves more than 2^27 8-byte
# elements delivers only part of the data (measured on MI355X, tools: tests/test_gpu_parity.py
# ::test_all_to_all_large_message).  The exchange is therefore issued in rounds of bounded size.
A2A_MAX_ELEMS = 1 << 26
# up to this many 8-byte words per rank the partitioned exchange packs counts, keys and values into one message
ONE_MESSAGE_MAX_ELEMS = 1 << 23


def all_to_all_v(dist, out, inp, out_splits, in_splits, group=None, max_elems=A2A_MAX_ELEMS, big=None):
    """all-to-all-v of 8-byte elements in rounds of at most max_elems per call; `out` keeps the
    (source rank, source order) layout of a single all_to_all_single.  big = largest chunk any pair of
    ranks exchanges, when the caller already knows it (else one all_reduce finds it)."""
    import torch
    P = len(in_splits)
    if big is None:
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

    # -- partitioned exchange (one pass = owner split + first partition pass of the local flush):
    #    see esp_shard_partition / esp_shard_assemble
    def part_partition(self, P, me, entries_per_shard):
        """-> None (not applicable) or (keys, vals, counts, entry_offsets, digits): device views of the
        partitioned pending entries, the per-(owner, digit) counts, owner r = entry_offsets[r:r+2]."""
        torch = self.torch
        d = self.A._d
        d.commit()
        ok = C.c_int32(0)
        pk, pv, pc = C.c_void_p(), C.c_void_p(), C.c_void_p()
        eoff = np.zeros(P + 1, np.int64)
        nb = C.c_int64(0)
        d.ck(d.lib.esp_shard_partition(d.h, P, me, int(entries_per_shard), C.byref(ok), C.byref(pk), C.byref(pv),
                                       C.byref(pc), eoff.ctypes.data_as(C.c_void_p), C.byref(nb)))
        self.A._touch()
        if not ok.value:
            return None
        n = int(eoff[-1])
        keys = _wrap_device(torch, pk.value, n, torch.int64, self.device)
        vals = _wrap_device(torch, pv.value, n, torch.float64, self.device)
        counts = _wrap_device(torch, pc.value, P * nb.value, torch.int64, self.device)
        return keys, vals, counts, eoff, int(nb.value)

    def part_wait(self):
        """The key/value views of part_partition are complete (the scatter kernel may still run when it returns)."""
        self.A.synchronize()

    def part_assemble(self, P, me, rkeys, rvals, rcounts, recv_entries):
        """rkeys/rvals/rcounts: per source rank a device tensor (entry `me` ignored).  The tensors are
        kept alive until the flush.  -> True: the flush runs on the pieces; False: plain pending buffer."""
        d = self.A._d
        if P > 1:
            self.torch.cuda.synchronize(self.device)  # the collectives ran on torch's stream
        PK, PV, PC = (C.c_void_p * P)(), (C.c_void_p * P)(), (C.c_void_p * P)()
        for q in range(P):
            if q != me:
                PK[q] = rkeys[q].data_ptr() if rkeys[q].numel() else None
                PV[q] = rvals[q].data_ptr() if rvals[q].numel() else None
                PC[q] = rcounts[q].data_ptr()
        ne = np.asarray(recv_entries, np.int64)
        ok = C.c_int32(0)
        d.ck(d.lib.esp_shard_assemble(d.h, PK, PV, PC, ne.ctypes.data_as(C.c_void_p), C.byref(ok)))
        self._alive = (rkeys, rvals, rcounts) if ok.value else None
        self.A._touch()
        return bool(ok.value)

    def flush(self):
        self.A.flush()
        self._alive = None
        return self.A._d.nnz()

    def local_csc(self):
        return self.A.sparse()


class ShardedExtendableSparseMatrix:
    """ExtendableSparseMatrix whose columns are sharded over the ranks of a process group."""

    def __init__(self, m, n, backend, group=None, dist=None, ctrl_group=None):
        """ctrl_group: a process group over the same ranks for the small host-side agreements of a flush (entry
        counts, "my stream is pre-sorted") -- a gloo group keeps them off the GPU, where a tiny collective queued
        behind the partition's scatter kernel only runs once that kernel has drained (measured: the device copy
        of three integers took the scatter kernel's 1.2 ms); None: the data group with device tensors."""
        if dist is None:   # (tests inject a stand-in that runs several ranks inside one process)
            import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.ctrl_group = ctrl_group
        self.rank = dist.get_rank(group)
        self.P = dist.get_world_size(group)
        self.m, self.n = int(m), int(n)
        self.backend = backend
        self.ranges = owner_ranges(self.n, self.P)
        self.local_nnz = 0
        self.nnz_offsets = np.zeros(self.P + 1, np.int64)
        self.partitioned = True      # try the partitioned exchange (falls back by consensus)
        self.last_exchange = None    # "partitioned" | "inplace" | "generic"
        self._eps = None             # entries per shard of the previous flush: fixes the digit width
        self._part_skip = 0          # back-off after a flush where some rank could not partition
        self._part_penalty = 0
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
        if self.partitioned and hasattr(be, "part_partition") and self._flush_exchange_partitioned():
            self.last_exchange = "partitioned"
        elif hasattr(be, "exchange_begin"):
            self.last_exchange = "inplace"
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
            self.last_exchange = "generic"
            self._flush_exchange_generic()
        self.local_nnz = be.flush()
        self._offsets_valid = False   # gathered on first use (nnz / local_slice / gather_sparse), see _offsets
        return self

    def _offsets(self):
        """Global colptr offsets = exclusive scan of the per-shard nnz.  COLLECTIVE on first use after a
        flush (an all_gather): like flush itself, every rank has to get here."""
        if not getattr(self, "_offsets_valid", True):
            counts = self._gather_ints([self.local_nnz])[:, 0]
            self.nnz_offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
            self._offsets_valid = True
        return self.nnz_offsets


    def _gather_ints(self, values):
        """all_gather of a small int64 vector -> (P, len) numpy array (same on every rank)."""
        import torch
        if self.ctrl_group is not None:   # host tensors over the control group: nothing is queued on the GPU
            mine = torch.tensor(list(values), dtype=torch.int64)
            out = [torch.empty_like(mine) for _ in range(self.P)]
            self.dist.all_gather(out, mine, group=self.ctrl_group)
            return torch.stack(out).numpy()
        dev = getattr(self.backend, "device", None)
        mine = torch.tensor(list(values), dtype=torch.int64, device=dev if dev is not None else "cpu")
        out = [torch.empty_like(mine) for _ in range(self.P)]
        self.dist.all_gather(out, mine, group=self.group)
        return torch.stack(out).cpu().numpy()   # one device-to-host copy, one synchronisation

    def _flush_exchange_partitioned(self):
        """One partition pass per rank (owner split + first pass of the local flush), ranges and
        per-digit counts to the owners, pieces assembled without a copy.  Every decision that changes
        the communication pattern is taken from all-gathered data, i.e. identically on all ranks.
        Returns False when the ranks agreed to use the plain exchange for this flush."""
        import torch
        dist, P, be, me = self.dist, self.P, self.backend, self.rank
        if self._part_skip > 0:
            self._part_skip -= 1
            return False
        if self._eps is None:   # first flush: the digit width comes from the global number of entries
            self._eps = -(-int(self._gather_ints([be.pending()]).sum()) // P)
        part = be.part_partition(P, me, self._eps)
        counts = np.diff(part[3]) if part is not None else np.zeros(P, np.int64)
        M = self._gather_ints([1 if part is not None else 0] + [int(c) for c in counts])
        total = int(M[:, 1:].sum())
        if not M[:, 0].all():
            # some rank's stream is not pre-sorted (or the plan does not apply): plain exchange now, and
            # for the next few flushes (the pending entries are intact)
            self._part_penalty = min(16, 2 * self._part_penalty + 1)
            self._part_skip = self._part_penalty
            self._eps = None
            return False
        self._part_penalty = 0
        self._eps = -(-total // P) if total else None
        # (the consensus round above and the host work below run beside the partition's scatter kernel:
        # be.part_wait() comes right before the first operation that reads the moved entries)
        keys, vals, cnts, eoff, nb = part
        in_x = [int(c) for c in counts]
        in_x[me] = 0
        out_x = [int(M[q, 1 + me]) for q in range(P)]
        out_x[me] = 0
        own_lo, own_hi = int(eoff[me]), int(eoff[me + 1])
        pairs = M[:, 1:].copy()
        np.fill_diagonal(pairs, 0)
        big = int(pairs.max()) if P > 1 else 0
        # what the busiest rank sends or receives, in 8-byte words (same number on every rank)
        busiest = int(max(pairs.sum(axis=1).max(), pairs.sum(axis=0).max())) * 2 + (P - 1) * nb if P > 1 else 0
        if 1 < P and busiest <= ONE_MESSAGE_MAX_ELEMS and 2 * big + nb <= A2A_MAX_ELEMS // P:
            # small exchange (a slab-wise assembly): counts, keys and values of one destination travel as ONE
            # message [nb counts | keys | values]: one collective instead of three
            parts, in_f = [], []
            for r in range(P):
                if r == me:
                    in_f.append(0)
                    continue
                lo, hi = int(eoff[r]), int(eoff[r + 1])
                parts += [cnts[r * nb:(r + 1) * nb], keys[lo:hi], vals[lo:hi].view(torch.int64)]
                in_f.append(nb + 2 * (hi - lo))
            out_f = [0 if q == me else nb + 2 * out_x[q] for q in range(P)]
            rbuf = be.empty(sum(out_f), torch.int64)
            be.part_wait()
            sbuf = torch.cat(parts)
            dist.all_to_all_single(rbuf, sbuf, out_f, in_f, group=self.group)
            fo = np.concatenate([[0], np.cumsum(out_f)]).astype(np.int64)
            rk, rv, rc = [], [], []
            for q in range(P):
                o, c = int(fo[q]), out_x[q]
                if q == me:
                    rk.append(rbuf[:0]), rv.append(rbuf[:0].view(torch.float64)), rc.append(rbuf[:0])
                else:
                    rc.append(rbuf[o:o + nb])
                    rk.append(rbuf[o + nb:o + nb + c])
                    rv.append(rbuf[o + nb + c:o + nb + 2 * c].view(torch.float64))
            be.part_assemble(P, me, rk, rv, rc, out_x)
            self.last_messages = 1
        else:
            rkeys = be.empty(sum(out_x), torch.int64)
            rvals = be.empty(sum(out_x), torch.float64)
            rcnts = be.empty(P * nb, torch.int64)
            ro = np.concatenate([[0], np.cumsum(out_x)]).astype(np.int64)
            if P > 1:   # (a single shard hands its pieces to the library on the same stream: no wait needed)
                be.part_wait()
            if sum(in_x):
                skeys = torch.cat([keys[:own_lo], keys[own_hi:]])
                svals = torch.cat([vals[:own_lo], vals[own_hi:]])
            else:
                skeys, svals = keys[:0], vals[:0]
            if P > 1:
                all_to_all_v(dist, rkeys, skeys, out_x, in_x, self.group, big=big)
                all_to_all_v(dist, rvals, svals, out_x, in_x, self.group, big=big)
                dist.all_to_all_single(rcnts, cnts, group=self.group)   # nb counts to / from every rank
            be.part_assemble(P, me,
                             [rkeys[ro[q]:ro[q + 1]] for q in range(P)],
                             [rvals[ro[q]:ro[q + 1]] for q in range(P)],
                             [rcnts[q * nb:(q + 1) * nb] for q in range(P)], out_x)
            self.last_messages = 3
        self.exchanged = (int(sum(counts)), int(sum(out_x)) + int(counts[me]))
        self.sent_off_rank = int(sum(in_x))
        return True

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
        return int(self._offsets()[-1])

    def local_slice(self):
        """This shard's part of the global CSC: (c0, c1, colptr[c0..c1] global 1-based, rowval, nzval)."""
        csc = self.backend.local_csc()
        c0, c1 = self.ranges[self.rank]
        colptr = csc.colptr[c0:c1 + 1] + self._offsets()[self.rank]
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
        colptr[-1] = self._offsets()[-1] + 1
        return SparseMatrixCSC(self.m, self.n, colptr, np.concatenate(rows), np.concatenate(vals))


