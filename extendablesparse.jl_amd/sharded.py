"""Column-range sharded assembly across the GPUs of one node: the Python caller of the C group
API.  One process per GPU; every rank appends whatever its part of the assembly loop produces -- like one reference
buffer per `tid` (src/matrix/genericmtextendablesparsematrixcsc.jl:87-99) -- and flush! is ONE collective C call per
rank (esp_group_flush): the exchange policy (csrc/group_policy.hpp) and the RCCL transport over xGMI (csrc/group.hpp)
live inside libesparse_hip.so.

    owner(col) = floor((col-1) * P / n)                      contiguous column ranges

(The Python orchestration of the same building blocks over torch.distributed, which the CPU / gloo tests run, is test
infrastructure: tests/sharded_model.py.)
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


class GroupShardedMatrix:
    """The sharded matrix through the C group API (esp_group_*): the exchange (RCCL all-to-all-v over xGMI, grouped
    ncclSend/ncclRecv on the handle's stream) and its policy live inside libesparse_hip.so; this class only holds the
    handles.  What a Julia / MPI host does with the same six calls.

    unique_id: the 128 bytes of GroupShardedMatrix.unique_id() made on rank 0 and broadcast by the host;
    comm: an esp_comm_t callback table instead (a host with its own transport; the ranks-as-threads tests)."""

    def __init__(self, m, n, nranks=1, rank=0, device=0, capacity_hint=0, unique_id=None, comm=None):
        self.m, self.n, self.P, self.rank = int(m), int(n), int(nranks), int(rank)
        self.A = ExtendableSparseMatrix(m, n, device=device, capacity_hint=capacity_hint)
        d = self.A._d
        self._g = C.c_void_p()
        self._comm = comm   # (keeps the callbacks alive)
        if comm is not None:
            d.ck(d.lib.esp_group_create_comm(d.h, self.P, self.rank, C.byref(comm), C.byref(self._g)))
        else:
            if unique_id is None:
                if self.P != 1:
                    raise ValueError("unique_id: rank 0 makes it with GroupShardedMatrix.unique_id(), the host broadcasts it")
                unique_id = self.unique_id()
            buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
            d.ck(d.lib.esp_group_create(d.h, self.P, self.rank, buf, C.byref(self._g)))
        self.local_nnz = 0

    @staticmethod
    def unique_id():
        lib = L.load()
        buf = (C.c_uint8 * 128)()
        L.check(None, lib.esp_group_unique_id(buf))
        return bytes(buf)

    def __del__(self):
        try:
            if self._g:
                self.A._d.lib.esp_group_destroy(self._g)
                self._g = C.c_void_p()
        except Exception:
            pass

    @property
    def local(self):
        return self.A

    matrix = local

    def debug_loopback(self, on=True):
        """Test hook (one rank, RCCL transport): every flush sends the own ranges to this very rank through the library's
        all-to-all-v and restores them from what arrived (include/esparse_hip.h, esp_debug_group_loopback)."""
        self._ck(self.A._d.lib.esp_debug_group_loopback(self._g, 1 if on else 0, None))

    def loopback_bytes(self):
        b = C.c_int64()
        self._ck(self.A._d.lib.esp_debug_group_loopback(self._g, -1, C.byref(b)))
        return b.value

    def _ck(self, rc):
        if rc != 0:
            msg = self.A._d.lib.esp_group_last_error(self._g)
            raise L.EspError(rc, msg.decode() if msg else "")

    def flush(self):
        d = self.A._d
        d.commit()
        z, ch = C.c_int64(), C.c_int32()
        self._ck(d.lib.esp_group_flush(self._g, ESP_FLUSH_ROUTED, C.byref(z), C.byref(ch)))
        self.A._touch()
        if ch.value:
            self.A._phash = None
        self.local_nnz = z.value
        return self

    def nnz(self):
        tot, before = C.c_int64(), C.c_int64()
        self._ck(self.A._d.lib.esp_group_nnz(self._g, C.byref(tot), C.byref(before)))
        return tot.value

    def column_range(self):
        lo, hi = C.c_int64(), C.c_int64()
        self._ck(self.A._d.lib.esp_group_column_range(self._g, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    @property
    def last_exchange(self):
        k, s = C.c_int32(), C.c_int64()
        self._ck(self.A._d.lib.esp_group_last_exchange(self._g, C.byref(k), C.byref(s)))
        return {1: "partitioned", 2: "inplace"}.get(k.value)

    @property
    def sent_off_rank(self):
        k, s = C.c_int32(), C.c_int64()
        self._ck(self.A._d.lib.esp_group_last_exchange(self._g, C.byref(k), C.byref(s)))
        return s.value

    def local_slice(self):
        """This shard's part of the global CSC: (c0, c1, colptr[c0..c1] global 1-based, rowval, nzval); collective."""
        lo, hi = self.column_range()
        cp = np.empty(hi - lo + 2, np.int64)
        rv = np.empty(self.local_nnz, np.int64)
        nz = np.empty(self.local_nnz, np.float64)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        self._ck(self.A._d.lib.esp_group_get_csc(self._g, vp(cp), vp(rv), vp(nz)))
        return lo - 1, hi, cp, rv, nz

    @staticmethod
    def stitch(m, n, pieces, total_nnz):
        """Global SparseMatrixCSC from the ranks' local_slice() pieces (checks / small sizes)."""
        colptr = np.ones(n + 1, np.int64)
        rows, vals = [], []
        for (c0, c1, cp, rv, nz) in pieces:
            colptr[c0:c1 + 1] = cp
            rows.append(rv)
            vals.append(nz)
        colptr[-1] = total_nnz + 1
        return SparseMatrixCSC(m, n, colptr, np.concatenate(rows), np.concatenate(vals))
