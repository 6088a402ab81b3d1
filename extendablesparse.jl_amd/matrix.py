"""Host-side mirror of the reference's operator interface for the assembly path.

Same names, argument meaning and error behaviour as ExtendableSparse.jl (paths relative
to the reference repository), backed by libesparse_hip.so through the C ABI only:

  SparseMatrixCSC                  Julia's SparseArrays.SparseMatrixCSC{Float64,Int64} (host container)
  SparseMatrixHIPCOO               the extension buffer replacing SparseMatrixLNK
                                   (plugin contract: src/matrix/abstractsparsematrixextension.jl:6-14)
  ExtendableSparseMatrix           src/matrix/extendable.jl with the buffer AND the CSC device-resident
  GenericExtendableSparseMatrixCSC src/matrix/genericextendablesparsematrixcsc.jl (host CSC + buffer)
  GenericMTExtendableSparseMatrixCSC src/matrix/genericmtextendablesparsematrixcsc.jl (one buffer per tid)

Julia is not available in this image, so this Python layer plays the role the Julia shim of
INTEGRATION.md plays in production; it contains no arithmetic of its own.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from ._lib import (ESP_COO, ESP_FLUSH_PLUS, ESP_FLUSH_ROUTED, ESP_OP_ADD, ESP_OP_SUB, ESP_RAWUPDATE, ESP_SET,
                   ESP_UPDATE, BoundsError, check)

_OPS = {"+": ESP_OP_ADD, "-": ESP_OP_SUB, ESP_OP_ADD: ESP_OP_ADD, ESP_OP_SUB: ESP_OP_SUB}
try:  # operator.add / operator.sub are accepted like Julia's `+` / `-`
    import operator as _operator
    _OPS[_operator.add] = ESP_OP_ADD
    _OPS[_operator.sub] = ESP_OP_SUB
except Exception:  # pragma: no cover
    pass


def _op(op):
    try:
        return _OPS[op]
    except (KeyError, TypeError):
        raise NotImplementedError("op %r is not supported by the device buffer (only + and -); "
                                  "use the CPU SparseMatrixLNK path for it" % (op,))


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


class SparseMatrixCSC:
    """Host CSC container with Julia's layout: Int64 1-based colptr (n+1) / rowval, Float64 nzval."""

    def __init__(self, m, n, colptr=None, rowval=None, nzval=None):
        self.m, self.n = int(m), int(n)
        if colptr is None:  # spzeros(m,n)
            colptr = np.ones(self.n + 1, np.int64)
            rowval = np.empty(0, np.int64)
            nzval = np.empty(0, np.float64)
        self.colptr = np.ascontiguousarray(colptr, np.int64)
        self.rowval = np.ascontiguousarray(rowval, np.int64)
        self.nzval = np.ascontiguousarray(nzval, np.float64)

    @property
    def shape(self):
        return (self.m, self.n)

    def nnz(self):
        return int(self.colptr[-1] - 1)

    def findindex(self, i, j):
        """findindex(csc,i,j): src/matrix/sparsematrixcsc.jl:7-23 (1-based position or 0)."""
        if not (1 <= i <= self.m and 1 <= j <= self.n):
            raise BoundsError()
        r1, r2 = int(self.colptr[j - 1]), int(self.colptr[j] - 1)
        if r1 > r2:
            return 0
        k = r1 + int(np.searchsorted(self.rowval[r1 - 1:r2], i, side="left"))
        if k > r2 or self.rowval[k - 1] != i:
            return 0
        return k

    def __getitem__(self, ij):
        k = self.findindex(*ij)
        return float(self.nzval[k - 1]) if k else 0.0

    def arrays(self):
        return self.colptr, self.rowval, self.nzval

    def findnz(self):
        J = np.repeat(np.arange(1, self.n + 1, dtype=np.int64), np.diff(self.colptr))
        return self.rowval.copy(), J, self.nzval.copy()

    def copy(self):
        return SparseMatrixCSC(self.m, self.n, self.colptr.copy(), self.rowval.copy(), self.nzval.copy())

    def pattern_equal(self, other):
        """pattern_equal: src/matrix/sparsematrixcsc.jl:83-85."""
        return np.array_equal(self.colptr, other.colptr) and np.array_equal(self.rowval, other.rowval)

    def __eq__(self, other):
        return (isinstance(other, SparseMatrixCSC) and self.shape == other.shape and self.pattern_equal(other)
                and np.array_equal(self.nzval, other.nzval))

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csc_matrix((self.nzval, self.rowval - 1, self.colptr - 1), shape=self.shape)


class _Handle:
    """Owns one esp_handle (one device, one stream) and its pinned staging chunk."""

    def __init__(self, m, n, device=0, capacity_hint=0):
        self.lib = L.load()
        self.m, self.n = int(m), int(n)
        h = C.c_void_p()
        check(None, self.lib.esp_create(self.m, self.n, device, capacity_hint, C.byref(h)))
        self.h = h
        self._st = None
        self._nst = 0
        self._held = []   # device arrays the library reads at FLUSH time (esp_append_elements[_again]): kept alive until then

    def clone(self):
        """A second handle with the same CSC and pending entries (device-to-device): Base.copy."""
        self.commit()
        c = object.__new__(_Handle)
        c.lib, c.m, c.n = self.lib, self.m, self.n
        h = C.c_void_p()
        self.ck(self.lib.esp_clone(self.h, C.byref(h)))
        c.h, c._st, c._nst, c._held = h, None, 0, list(self._held)
        return c

    def close(self):
        if getattr(self, "h", None):
            self.lib.esp_destroy(self.h)
            self.h = None
            self._held = []

    def hold(self, *objs):
        """The device form of an element-level append reads elmat / diag (and, without cell records, cellnodes) when the batch is
        FLUSHED (include/esparse_hip.h: valid and unchanged until the handle's next esp_flush / esp_reset / esp_clear_pending has
        returned): the handle keeps the caller's array objects alive that long -- a temporary tensor must not go back to its
        allocator in between.  (Keeping them UNCHANGED stays the caller's part.)"""
        self._held.extend(o for o in objs if o is not None)

    def drop_held(self):
        self._held = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def ck(self, rc):
        check(self.h, rc)

    # -- staged per-entry appends (one ccall per chunk, not per entry)
    def _stage(self):
        if self._st is None:
            r, c, v, k = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
            got = C.c_int64()
            self.ck(self.lib.esp_stage_begin(self.h, 1 << 16, C.byref(r), C.byref(c), C.byref(v), C.byref(k),
                                             C.byref(got)))
            n = got.value
            self._st = (np.ctypeslib.as_array(C.cast(r, C.POINTER(C.c_int64)), (n,)),
                        np.ctypeslib.as_array(C.cast(c, C.POINTER(C.c_int64)), (n,)),
                        np.ctypeslib.as_array(C.cast(v, C.POINTER(C.c_double)), (n,)),
                        np.ctypeslib.as_array(C.cast(k, C.POINTER(C.c_uint8)), (n,)))
        return self._st

    def push(self, kind, v, i, j):
        if not (1 <= i <= self.m and 1 <= j <= self.n):
            raise BoundsError("(%d,%d) outside %d x %d" % (i, j, self.m, self.n))
        r, c, vals, k = self._stage()
        n = self._nst
        r[n], c[n], vals[n], k[n] = i, j, v, kind
        self._nst = n + 1
        if self._nst == len(r):
            self.commit()

    def commit(self):
        if self._nst:
            n, self._nst = self._nst, 0
            self.ck(self.lib.esp_commit(self.h, n, -1, ESP_OP_ADD))

    def append(self, kind, I, J, V, op=ESP_OP_ADD, kinds=None):
        self.commit()
        i32 = getattr(I, "dtype", None) == np.int32 and getattr(J, "dtype", None) == np.int32   # (Ti = Int32 arrays go as they are)
        I = np.ascontiguousarray(I, np.int32 if i32 else np.int64)
        J = np.ascontiguousarray(J, np.int32 if i32 else np.int64)
        V = np.ascontiguousarray(V, np.float64)
        kp = None
        if kinds is not None:
            kinds = np.ascontiguousarray(kinds, np.uint8)
            kp = _vp(kinds)
        fn = self.lib.esp_append_host_i32 if i32 else self.lib.esp_append_host
        self.ck(fn(self.h, _vp(I), _vp(J), _vp(V), kp, kind, op, len(I)))

    def append_device(self, kind, I, J, V, op=ESP_OP_ADD, kinds=None):
        """esp_append_device: I, J (int64), V (float64) [, kinds (uint8)] are arrays resident in this GPU's memory --
        anything with .data_ptr() and .numel() (torch tensors); nothing is staged through the host."""
        self.commit()
        n = int(I.numel())
        if int(J.numel()) != n or int(V.numel()) != n or (kinds is not None and int(kinds.numel()) != n):
            raise ValueError("append_device: arrays of different lengths")
        kp = C.c_void_p(kinds.data_ptr()) if kinds is not None else None
        self.ck(self.lib.esp_append_device(self.h, C.c_void_p(I.data_ptr()), C.c_void_p(J.data_ptr()), C.c_void_p(V.data_ptr()), kp, kind, op, n))

    def append_elements(self, kind, cellnodes, elmat, diag=None, op=ESP_OP_ADD):
        """esp_append_elements[_host]: the inner loops of test/femtools.jl:61-69 for element data in arrays -- cellnodes
        (nloc x ncells, int64), elmat (nloc x nloc x ncells, float64), diag (nloc x ncells) or None, all in Julia's
        (column-major) layout: NumPy arrays in Fortran order (host), or anything with .data_ptr() / .numel() resident on this
        GPU whose memory is laid out that way (a torch tensor of shape (ncells, nloc[, nloc]) -- its C order IS that layout,
        elmat[c, jl, il]).  Device arrays are read at FLUSH time and must stay unchanged until then (the handle keeps them alive: hold)."""
        self.commit()
        if hasattr(cellnodes, "data_ptr"):
            nloc = int(round(elmat.numel() / max(cellnodes.numel(), 1)))
            if nloc < 1 or cellnodes.numel() % nloc or elmat.numel() != cellnodes.numel() * nloc or \
                    (diag is not None and diag.numel() != cellnodes.numel()):
                raise ValueError("append_elements: array sizes do not fit together")
            nc = cellnodes.numel() // nloc
            dp = C.c_void_p(diag.data_ptr()) if diag is not None else None
            self.hold(cellnodes, elmat, diag)
            self.ck(self.lib.esp_append_elements(self.h, nloc, nc, C.c_void_p(cellnodes.data_ptr()), C.c_void_p(elmat.data_ptr()), dp, kind, op))
            return
        cn = np.asfortranarray(cellnodes, np.int64)
        nloc, nc = cn.shape
        em = np.asfortranarray(elmat, np.float64)
        if em.shape != (nloc, nloc, nc):
            raise ValueError("append_elements: elmat must be nloc x nloc x ncells")
        dg = None
        if diag is not None:
            dg = np.asfortranarray(diag, np.float64)
            if dg.shape != (nloc, nc):
                raise ValueError("append_elements: diag must be nloc x ncells")
        self.ck(self.lib.esp_append_elements_host(self.h, nloc, nc, _vp(cn), _vp(em), _vp(dg) if dg is not None else None, kind, op))

    def pending(self):
        c = C.c_int64()
        self.ck(self.lib.esp_pending(self.h, C.byref(c)))
        return c.value + self._nst

    def nnz(self):
        c = C.c_int64()
        self.ck(self.lib.esp_nnz(self.h, C.byref(c)))
        return c.value

    def set_csc(self, csc):
        self.ck(self.lib.esp_set_csc(self.h, _vp(csc.colptr), _vp(csc.rowval), _vp(csc.nzval), csc.nnz()))

    def flush(self, mode):
        self.commit()
        z, ch = C.c_int64(), C.c_int32()
        self.ck(self.lib.esp_flush(self.h, mode, C.byref(z), C.byref(ch)))
        self.drop_held()
        return z.value, bool(ch.value)

    def get_csc(self):
        z = self.nnz()
        cp = np.empty(self.n + 1, np.int64)
        rv = np.empty(z, np.int64)
        nz = np.empty(z, np.float64)
        self.ck(self.lib.esp_get_csc(self.h, _vp(cp), _vp(rv), _vp(nz)))
        return SparseMatrixCSC(self.m, self.n, cp, rv, nz)

    def timing(self, clear=True):
        t = L.esp_timing_t()
        self.ck(self.lib.esp_timing(self.h, C.byref(t), 1 if clear else 0))
        d = {name: (t.ms[i], t.launches[i]) for i, name in enumerate(L.STAGES)}
        d["flush_ms"], d["flushes"] = t.flush_ms, t.flushes
        return d


# False: every upload sends the whole matrix (esp_set_csc).  The values-only path recognises "the pattern the device already holds"
# by the IDENTITY of the colptr / rowval arrays the last download handed out: an in-place edit of those arrays that keeps their
# length goes unseen (the device would keep the old pattern).  The contract -- as for Julia's SparseMatrixCSC, whose
# constructor the reference calls with fresh vectors -- is: assign a new matrix, never edit colptr / rowval in place; a caller
# that cannot promise that sets this switch.
VALUES_ONLY_UPLOAD = True


def _upload_csc(d, mirror, csc):
    """Bring the device CSC of handle `d` to `csc`.  `mirror` = the (colptr, rowval) array objects of the host matrix the
    device CSC equals in pattern (what the last download handed out): when `csc` still carries those very arrays only the
    values travel (esp_set_nzval: 8 instead of 24 bytes per entry) -- the Generic wrappers edit cscmatrix.nzval in place
    (genericextendablesparsematrixcsc.jl:44-54), which nobody can see from outside, so the values always do.
    colptr / rowval must never be edited in place (VALUES_ONLY_UPLOAD above)."""
    if VALUES_ONLY_UPLOAD and mirror is not None and mirror[0] is csc.colptr and mirror[1] is csc.rowval and d.nnz() == csc.nnz():
        d.ck(d.lib.esp_set_nzval(d.h, _vp(np.ascontiguousarray(csc.nzval, np.float64))))
    else:
        d.set_csc(csc)


def _download_csc(d, csc, changed):
    """The device CSC of `d` as a host SparseMatrixCSC; the pattern arrays of `csc` are shared when the flush kept the
    pattern (esp_get_nzval: values only).  Returns (matrix, mirror)."""
    if not changed and d.nnz() == csc.nnz():
        nz = np.empty(csc.nnz(), np.float64)
        d.ck(d.lib.esp_get_nzval(d.h, _vp(nz)))
        out = SparseMatrixCSC(d.m, d.n, csc.colptr, csc.rowval, nz)
    else:
        out = d.get_csc()
    return out, (out.colptr, out.rowval)


class SparseMatrixHIPCOO:
    """Device-resident COO append buffer: the `T_ext` of the reference's plugin contract
    (abstractsparsematrixextension.jl:6-14), replacing SparseMatrixLNK (sparsematrixlnk.jl)."""

    def __init__(self, m, n, device=0, capacity_hint=0):
        self._d = _Handle(m, n, device, capacity_hint)
        self.m, self.n = int(m), int(n)
        self._device = device
        self._mirror = None   # (colptr, rowval) of the host matrix whose pattern the handle's device CSC holds

    def size(self):
        return (self.m, self.n)

    shape = property(size)

    def nnz(self):
        """>0 iff anything is pending (the flush! gate, genericextendablesparsematrixcsc.jl:32)."""
        return self._d.pending()

    def __setitem__(self, ij, v):  # setindex!: sparsematrixlnk.jl:178-201
        self._d.push(ESP_SET, float(v), int(ij[0]), int(ij[1]))

    def __getitem__(self, ij):
        """getindex(buffer,i,j) (sparsematrixlnk.jl:151-171): what the pending entries alone leave at (i,j) -- their
        ordered fold on the device, 0.0 if no entry exists.  One pass over the pending keys per call (slow path)."""
        i, j = int(ij[0]), int(ij[1])
        if not (1 <= i <= self.m and 1 <= j <= self.n):
            raise BoundsError("(%d,%d) outside %d x %d" % (i, j, self.m, self.n))
        self._d.commit()
        val, found = C.c_double(), C.c_int32()
        self._d.ck(self._d.lib.esp_pending_getindex(self._d.h, i, j, C.byref(val), C.byref(found)))
        return val.value

    def release(self):
        """Free the device and pinned memory of this buffer now (what the shim does with the buffer a Generic
        wrapper drops after flush!: the garbage collector does not see device memory)."""
        self._d._st, self._d._nst = None, 0
        self._d.ck(self._d.lib.esp_release_buffers(self._d.h))
        self._d.drop_held()

    def updateindex(self, op, v, i, j):  # updateindex!: sparsematrixlnk.jl:210-228
        v = float(v)
        self._d.push(ESP_UPDATE, -v if _op(op) == ESP_OP_SUB else v, int(i), int(j))

    def rawupdateindex(self, op, v, i, j, tid=1):  # rawupdateindex!: sparsematrixlnk.jl:237-253
        v = float(v)
        self._d.push(ESP_RAWUPDATE, -v if _op(op) == ESP_OP_SUB else v, int(i), int(j))

    def append(self, kind, I, J, V, op="+", kinds=None):
        """Bulk form of the three calls above (one C call for the whole batch)."""
        self._d.append(kind, I, J, V, _op(op), kinds)

    def append_elements(self, cellnodes, elmat, diag=None, kind=ESP_RAWUPDATE, op="+"):
        """The assembly loop of test/femtools.jl:61-69 over element arrays, into this buffer (see _Handle.append_elements)."""
        self._d.append_elements(kind, cellnodes, elmat, diag, _op(op))

    def __add__(self, csc):
        """Base.:+(ext, csc) -> SparseMatrixCSC (sparsematrixlnk.jl:294-383): THE flush.  The handle keeps the result on
        the device: the next `+` with the matrix this one returned uploads values only, and a flush that adds no new
        position downloads values only."""
        if (csc.m, csc.n) != (self.m, self.n):
            raise AssertionError("size mismatch")
        self._d.commit()
        _upload_csc(self._d, self._mirror, csc)
        _, changed = self._d.flush(ESP_FLUSH_PLUS)
        out, self._mirror = _download_csc(self._d, csc, changed)
        return out

    __radd__ = __add__

    @staticmethod
    def sum(xs, csc, home=None):
        """Base.sum(extmatrices, csc) (sparsematrixdilnkc.jl:397-435): ((csc + x1) + x2) + ... left to right, as ONE device
        call (esp_flush_sum): every buffer folds by itself, the folds meet the stored matrix in one flush, the CSC
        travels once -- values only when `home` (a SparseMatrixHIPCOO kept by the caller between flushes) still holds
        its pattern.  The buffers come back empty."""
        for x in xs:
            x._d.commit()
        if sum(x._d.pending() for x in xs) == 0:
            return csc
        dst = home if home is not None else SparseMatrixHIPCOO(csc.m, csc.n, device=xs[0]._device)
        d = dst._d
        _upload_csc(d, dst._mirror, csc)
        arr = (C.c_void_p * len(xs))(*[x._d.h for x in xs])
        z, ch = C.c_int64(), C.c_int32()
        d.ck(d.lib.esp_flush_sum(d.h, arr, len(xs), C.byref(z), C.byref(ch)))
        for x in xs:
            x._d.drop_held()          # (the buffers come back empty)
        out, dst._mirror = _download_csc(d, csc, bool(ch.value))
        if home is None:
            d.close()
        return out


class ExtendableSparseMatrix:
    """ExtendableSparseMatrixCSC{Float64,Int64} (src/matrix/extendable.jl) with the extension buffer
    AND the CSC resident on the GPU.  Every update is appended to the device buffer; flush! runs
    the HIP pipeline in ROUTED mode, which applies updates of entries already in the CSC in call
    order (extendable.jl:164-166) and merges the rest (extendable.jl:248-255)."""

    # state of the host copy behind the `cscmatrix` property (ESparseHIP.jl: HOST_CURRENT / HOST_VALUES_STALE / HOST_STALE)
    HOST_CURRENT, HOST_VALUES_STALE, HOST_STALE = 0, 1, 2

    def __init__(self, m, n=None, device=0, capacity_hint=0, host_edits=True):
        if isinstance(m, SparseMatrixCSC):  # extendable.jl:61-63
            csc = m
            self._d = _Handle(csc.m, csc.n, device, capacity_hint)
            self._d.set_csc(csc)
            self.m, self.n = csc.m, csc.n
            self._phash = None  # phash(csc), evaluated on first use
        else:
            self._d = _Handle(m, n, device, capacity_hint)
            self.m, self.n = int(m), int(n)
            self._phash = 0  # extendable.jl:40
        self._host = None
        self._host_state = self.HOST_STALE
        self._handed_out = False      # read through `cscmatrix` since the last upload: its nzval may carry host edits
        self.host_edits = host_edits  # False: the caller promises never to edit the host copy (no esp_set_nzval)

    @classmethod
    def from_coo(cls, I, J, V, m=None, n=None, combine="+", device=0):
        """ExtendableSparseMatrixCSC(I,J,V[,m,n][,combine]) (extendable.jl:85-104) = sparse(I,J,V,m,n,+):
        the triplets go through the same device pipeline as incremental updates, as COO entries
        (first value as it is, duplicates added in input order, numerical zeros kept).  Only
        combine = + runs on the device (every other function stays with SparseArrays on the CPU)."""
        if _op(combine) != ESP_OP_ADD:
            raise ValueError("combine: only + runs on the device")
        I = np.ascontiguousarray(I, np.int64)
        J = np.ascontiguousarray(J, np.int64)
        m = int(I.max()) if m is None else int(m)     # sparse(I,J,V) = sparse(I,J,V,maximum(I),maximum(J))
        n = int(J.max()) if n is None else int(n)
        A = cls(m, n, device=device, capacity_hint=len(I))
        A.append(ESP_COO, I, J, V)
        A.flush()
        return A

    @property
    def shape(self):
        return (self.m, self.n)

    size = shape

    @property
    def phash(self):
        """ext.phash (extendable.jl:24): recomputed after every structural flush! (:252).  The hash
        kernel runs on first use after such a flush instead of inside flush! (same value)."""
        if self._phash is None:
            self._phash = self._pattern_hash()
        return self._phash

    def _pattern_hash(self):
        hsh = C.c_uint64()
        self._d.ck(self._d.lib.esp_pattern_hash(self._d.h, C.byref(hsh)))
        return hsh.value

    def _push_edits(self):
        """nzval of a handed-out host copy back to the device (esp_set_nzval) in front of the next update, flush! or device
        consumer: the reference's callers edit ext.cscmatrix.nzval in place (nonzeros(A) .= 0: sprand.jl:82,
        test_parallel.jl:71-92).  Same pattern on both sides: the copy was current when it was handed out."""
        if self._handed_out:
            self._handed_out = False
            if self.host_edits and self._host_state == self.HOST_CURRENT and self._host is not None and self._host.nnz() > 0:
                self._d.ck(self._d.lib.esp_set_nzval(self._d.h, _vp(np.ascontiguousarray(self._host.nzval, np.float64))))

    def _touch(self, state=1):
        self._push_edits()
        self._host_state = max(self._host_state, state)

    @property
    def cscmatrix(self):
        """The FIELD ext.cscmatrix of extendable.jl:10-25 as the reference's consumers read it right after flush!
        (factorizations/ilu0.jl:126-136, umfpack_lu.jl:18-27, jacobi.jl:54-64): flushes, brings the host copy up to date --
        nothing travels when it is current, nzval only (esp_get_nzval, INTO the array handed out before) when no position was
        added since the last read, else the whole matrix (esp_get_csc) -- and returns a valid SparseMatrixCSC, never None."""
        self.flush()
        if self._host_state != self.HOST_CURRENT or self._host is None:
            d = self._d
            if self._host_state == self.HOST_VALUES_STALE and self._host is not None and d.nnz() == self._host.nnz():
                d.ck(d.lib.esp_get_nzval(d.h, _vp(self._host.nzval)))
            else:
                self._host = d.get_csc()
            self._host_state = self.HOST_CURRENT
        self._handed_out = True
        return self._host

    @cscmatrix.setter
    def cscmatrix(self, csc):
        """ext.cscmatrix = B (what reset! of the reference does to the field): B becomes the stored matrix (esp_set_csc)."""
        assert (csc.m, csc.n) == (self.m, self.n)
        self._d.commit()
        self._d.set_csc(csc)
        self._host, self._host_state, self._handed_out = csc, self.HOST_CURRENT, True
        self._phash = None

    def __setitem__(self, ij, v):  # extendable.jl:205-218
        self._touch()
        self._d.push(ESP_SET, float(v), int(ij[0]), int(ij[1]))

    def updateindex(self, op, v, i, j):  # extendable.jl:159-174
        self._touch()
        v = float(v)
        self._d.push(ESP_UPDATE, -v if _op(op) == ESP_OP_SUB else v, int(i), int(j))

    def rawupdateindex(self, op, v, i, j, part=1):  # extendable.jl:181-197
        self._touch()
        v = float(v)
        self._d.push(ESP_RAWUPDATE, -v if _op(op) == ESP_OP_SUB else v, int(i), int(j))

    def append(self, kind, I, J, V, op="+", kinds=None):
        self._touch()
        self._d.append(kind, I, J, V, _op(op), kinds)

    def append_device(self, kind, I, J, V, op="+", kinds=None):
        """The bulk form for triplets that already lie in this GPU's memory (torch tensors: int64, int64, float64)."""
        self._touch()
        self._d.append_device(kind, I, J, V, _op(op), kinds)

    def append_elements(self, cellnodes, elmat, diag=None, kind=ESP_RAWUPDATE, op="+"):
        """The assembly loop of test/femtools.jl:61-69 as one call: for every cell and local row il the optional
        diag[il, cell] on (i, i), then elmat[il, jl, cell] on (i, cellnodes[jl, cell]) for every jl -- bit-identical to
        the per-entry rawupdateindex! / updateindex! calls in that order (see _Handle.append_elements for the layouts)."""
        self._touch()
        self._d.append_elements(kind, cellnodes, elmat, diag, _op(op))

    def elements_keep_plan(self, on=True):
        """esp_elements_keep_plan: the next append_elements on the empty buffer keeps its plan (item order, cell records) for
        append_elements_again -- the same mesh, new element matrices."""
        self._d.ck(self._d.lib.esp_elements_keep_plan(self._d.h, 1 if on else 0))

    def append_elements_again(self, elmat, diag=None, kind=ESP_RAWUPDATE, op="+"):
        """esp_append_elements_again[_host]: the element loop over the connectivity of the last planned append_elements; device
        arrays (anything with .data_ptr()) or NumPy arrays, in append_elements' layouts."""
        self._touch()
        self._d.commit()
        if not hasattr(elmat, "data_ptr"):      # host arrays (Fortran order, like append_elements)
            em = np.asfortranarray(elmat, np.float64)
            dg = None if diag is None else np.asfortranarray(diag, np.float64)
            self._d.ck(self._d.lib.esp_append_elements_again_host(self._d.h, _vp(em), _vp(dg) if dg is not None else None, kind, _op(op)))
            return
        dp = C.c_void_p(diag.data_ptr()) if diag is not None else None
        self._d.hold(elmat, diag)     # (read at flush time, like append_elements' device arrays)
        self._d.ck(self._d.lib.esp_append_elements_again(self._d.h, C.c_void_p(elmat.data_ptr()), dp, kind, _op(op)))

    def generate_fem_mesh(self, dim, npd, cellnodes, elmat, diag=None, seed=0x5EED0004, order_mode=1, node_mode=0,
                          node_seed=0x5EED0014, cell_begin=0, cell_end=None):
        """esp_generate_fem_mesh: fills DEVICE arrays (anything with .data_ptr()) with the element data of generate_fem's
        grid for the cells at stream positions [cell_begin, cell_end): what a caller of testassemble! would hold."""
        q = npd - 1
        nc = 2 * q * q if dim == 2 else 6 * q * q * q
        cell_end = nc if cell_end is None else cell_end
        dp = C.c_void_p(diag.data_ptr()) if diag is not None else None
        self._d.ck(self._d.lib.esp_generate_fem_mesh(self._d.h, dim, npd, seed, order_mode, node_mode, node_seed, cell_begin, cell_end,
                                                      C.c_void_p(cellnodes.data_ptr()), C.c_void_p(elmat.data_ptr()), dp))

    def __getitem__(self, ij):
        """getindex (extendable.jl:226-238).  Pending entries live on the device, so the lookup is
        flush-then-findindex: correct, slow, documented (use updateindex! instead of A[i,j]+=v)."""
        i, j = int(ij[0]), int(ij[1])
        if not (1 <= i <= self.m and 1 <= j <= self.n):
            raise BoundsError()
        self.flush()
        val, found = C.c_double(), C.c_int32()
        self._d.ck(self._d.lib.esp_getindex(self._d.h, i, j, C.byref(val), C.byref(found)))
        return val.value

    def flush(self):  # flush!: extendable.jl:248-255
        self._push_edits()
        if self._d.pending() > 0:
            self._touch()
            _, changed = self._d.flush(ESP_FLUSH_ROUTED)
            if changed:
                self._phash = None  # extendable.jl:252 (evaluated lazily by the phash property)
                self._host_state = self.HOST_STALE
        return self

    def sparse(self):  # extendable.jl:258-261: flush!, then the field
        return self.cscmatrix

    def nnz(self):  # abstractextendablesparsematrixcsc.jl:80
        self.flush()
        return self._d.nnz()

    def nnznew(self):
        return self._d.pending()

    def nonzeros(self):
        return self.sparse().nzval

    def rowvals(self):
        return self.sparse().rowval

    def getcolptr(self):
        return self.sparse().colptr

    def findnz(self):
        return self.sparse().findnz()

    def arrays(self):
        return self.sparse().arrays()

    def mul(self, x, out=None):
        """LinearAlgebra.mul!(r, ext, x) (abstractextendablesparsematrixcsc.jl:179-181): flush!, then r = A*x on
        the device CSC, every r[i] summed in increasing column order like the reference's column loop.
        x, out: NumPy arrays (copied through the device) or CUDA torch tensors (used in place)."""
        self.flush()
        d = self._d
        if hasattr(x, "is_cuda") and x.is_cuda:
            import torch
            assert x.dtype == torch.float64 and x.numel() == self.n and x.is_contiguous()
            r = out if out is not None else torch.empty(self.m, dtype=torch.float64, device=x.device)
            assert r.is_cuda and r.dtype == torch.float64 and r.numel() == self.m and r.is_contiguous()
            torch.cuda.current_stream(x.device).synchronize()   # the library runs on its own stream
            d.ck(d.lib.esp_mul(d.h, C.c_void_p(x.data_ptr()), C.c_void_p(r.data_ptr()), 1))
            return r
        x = np.ascontiguousarray(x, np.float64)
        if x.shape != (self.n,):
            raise ValueError("DimensionMismatch")
        r = out if out is not None else np.empty(self.m, np.float64)
        d.ck(d.lib.esp_mul(d.h, _vp(x), _vp(r), 0))
        return r

    def copy(self):
        """Base.copy(ext) (extendable.jl:279-285): CSC, pending entries and phash are copied."""
        self._push_edits()
        c = object.__new__(ExtendableSparseMatrix)
        c._d = self._d.clone()
        c.m, c.n = self.m, self.n
        c._phash = self._phash
        c._host, c._host_state, c._handed_out, c.host_edits = None, self.HOST_STALE, False, self.host_edits
        return c

    def mark_dirichlet(self, penalty=1.0e20):
        """mark_dirichlet(A; penalty) (sparsematrixcsc.jl:94-108, via abstractextendablesparsematrixcsc.jl): flush!,
        then the Bool vector marking the nodes with A[i,i] >= penalty."""
        self.flush()
        out = np.zeros(self.n, np.uint8)
        self._d.ck(self._d.lib.esp_mark_dirichlet(self._d.h, float(penalty), _vp(out), 0))
        return out.astype(bool)

    def eliminate_dirichlet(self, marker):
        """eliminate_dirichlet!(A, marker) (sparsematrixcsc.jl:121-144): A[:,i] = 0, A[i,:] = 0, A[i,i] = 1 for marked i."""
        self.flush()
        mk = np.ascontiguousarray(np.asarray(marker) != 0, np.uint8)
        if mk.shape != (self.n,):
            raise ValueError("DimensionMismatch")
        self._d.ck(self._d.lib.esp_eliminate_dirichlet(self._d.h, _vp(mk), 0))
        self._touch()
        return self

    def jacobi(self):
        """jacobi(A) (src/factorizations/jacobi.jl:5-12) on the device CSC: invdiag = 1 ./ diag(A)."""
        self.flush()
        out = np.empty(self.n, np.float64)
        self._d.ck(self._d.lib.esp_jacobi_setup(self._d.h, _vp(out), 0))
        return out

    def ilu0(self):
        """ilu0(A) (src/factorizations/ilu0.jl:8-41) on the device CSC: (xdiag, idiag)."""
        self.flush()
        xd = np.empty(self.n, np.float64)
        idg = np.empty(self.n, np.int64)
        self._d.ck(self._d.lib.esp_ilu0_setup(self._d.h, _vp(xd), _vp(idg), 0))
        return xd, idg

    def __matmul__(self, x):  # A*x (genericmtextendablesparsematrixcsc.jl:119-121)
        return self.mul(x)

    def reset(self):  # reset!: extendable.jl:269-272 (phash kept)
        self._handed_out = False
        self._host_state = self.HOST_STALE
        self._d._nst = 0
        self._d.ck(self._d.lib.esp_reset(self._d.h))
        self._d.drop_held()

    def zero_values(self):  # fdrand!'s zero!: sprand.jl:82
        self.flush()
        self._touch()
        self._d.ck(self._d.lib.esp_zero_values(self._d.h))

    def dropzeros(self):  # dropzeros!(ext): flush then dropzeros!(csc)
        self.flush()
        self._touch(self.HOST_STALE)
        z = C.c_int64()
        self._d.ck(self._d.lib.esp_dropzeros(self._d.h, C.byref(z)))
        return self

    def generate_fdrand(self, nx, ny=1, nz=1, seed=0x5EED0002, rand_mode=1, kind=ESP_UPDATE):
        """The hot loop of fdrand! (sprand.jl:100-124) produced on the device."""
        self._touch()
        self._d.commit()
        self._d.ck(self._d.lib.esp_generate_fdrand(self._d.h, nx, ny, nz, seed, rand_mode, kind))

    def generate_fdrand_range(self, nx, ny, nz, node_begin, node_end, seed=0x5EED0002, rand_mode=1, kind=ESP_UPDATE):
        """The updates issued by nodes [node_begin, node_end) (0-based) of the fdrand! loop nest."""
        self._touch()
        self._d.commit()
        self._d.ck(self._d.lib.esp_generate_fdrand_range(self._d.h, nx, ny, nz, seed, rand_mode, kind, node_begin, node_end))

    def set_column_window(self, col_lo, col_hi):
        self._d.ck(self._d.lib.esp_set_column_window(self._d.h, col_lo, col_hi))

    def generate_fem(self, dim, npd, seed=0x5EED0004, order_mode=1):
        """The update stream of testassemble! (test/femtools.jl:45-72) produced on the device."""
        self._touch()
        self._d.commit()
        self._d.ck(self._d.lib.esp_generate_fem(self._d.h, dim, npd, seed, order_mode))

    def debug_force_path(self, path):
        """Test hook: 0 automatic, 2 force the general path (global LSD sort + global fold); the other values select
        one implementation where the library has two (include/esparse_hip.h, esp_debug_force_path)."""
        self._d.ck(self._d.lib.esp_debug_force_path(self._d.h, path))

    def debug_plan_cap(self, cap):
        """Test hook (esp_debug_plan_cap): plan the partition as if the bucket kernel took segments of `cap` entries; 0: off."""
        self._d.ck(self._d.lib.esp_debug_plan_cap(self._d.h, float(cap)))

    def debug_fail_next_bucket_stage(self):
        """Test hook (esp_debug_fail_next_bucket_stage): the next flush fails at its bucket stage, once; the batch stays pending."""
        self._d.ck(self._d.lib.esp_debug_fail_next_bucket_stage(self._d.h))

    def debug_last_path(self):
        p = C.c_int32()
        self._d.ck(self._d.lib.esp_debug_last_path(self._d.h, C.byref(p)))
        return p.value

    def debug_last_local_small(self):
        """1: the last flush's bucket kernel was the small variant (three workgroups per CU)"""
        p = C.c_int32()
        self._d.ck(self._d.lib.esp_debug_last_local_small(self._d.h, C.byref(p)))
        return p.value

    def debug_last_lazy_items(self):
        """1: the last flush's bucket kernel formed its updates from sorted item records (the expansion never ran)"""
        p = C.c_int32()
        self._d.ck(self._d.lib.esp_debug_last_lazy_items(self._d.h, C.byref(p)))
        return p.value

    def debug_last_rebuild(self):
        """1: the last flush rebuilt the matrix for the entries behind a re-assembly's batch (stored entries as a first piece)"""
        p = C.c_int32()
        self._d.ck(self._d.lib.esp_debug_last_rebuild(self._d.h, C.byref(p)))
        return p.value

    def debug_last_shard_source(self):
        """1: the last esp_shard_partition moved the entries itself, 2: the producer had partitioned them"""
        p = C.c_int32()
        self._d.ck(self._d.lib.esp_debug_last_shard_source(self._d.h, C.byref(p)))
        return p.value

    def debug_last_partition(self):
        p = C.c_int32()
        self._d.ck(self._d.lib.esp_debug_last_partition(self._d.h, C.byref(p)))
        return p.value

    def debug_last_plan_reused(self):
        k = C.c_int32()
        self._d.ck(self._d.lib.esp_debug_last_plan_reused(self._d.h, C.byref(k)))
        return k.value

    def debug_last_run_order(self):
        """1 = ranking kernel, 2 = radix-ordered run list, 3 = ranking given up, radix-ordered list used"""
        p = C.c_int32()
        self._d.ck(self._d.lib.esp_debug_last_run_order(self._d.h, C.byref(p)))
        return p.value

    def debug_last_fold_update(self):
        """True when the bucket kernel of the last flush used its UPDATE-only fold"""
        p = C.c_int32()
        self._d.ck(self._d.lib.esp_debug_last_fold_update(self._d.h, C.byref(p)))
        return bool(p.value)

    def debug_last_key_bytes(self):
        """4: the bucket kernel of the last flush read 4-byte keys (one kind for all pending entries), else 8"""
        p = C.c_int32()
        self._d.ck(self._d.lib.esp_debug_last_key_bytes(self._d.h, C.byref(p)))
        return p.value

    def debug_last_colptr_direct(self):
        """True when the bucket kernel of the last flush wrote colptr itself (no scan over the columns)"""
        p = C.c_int32()
        self._d.ck(self._d.lib.esp_debug_last_colptr_direct(self._d.h, C.byref(p)))
        return bool(p.value)

    def timing_enable(self, on=True):
        """on = True/1: events around the big kernels; 2: around every stage (incl. the small scans); 3: around the
        bucket kernel (general path: fold kernel) only; False: off"""
        self._d.ck(self._d.lib.esp_timing_enable(self._d.h, int(on)))

    def timing(self, clear=True):
        return self._d.timing(clear)

    def synchronize(self):
        self._d.ck(self._d.lib.esp_synchronize(self._d.h))


class GenericExtendableSparseMatrixCSC:
    """src/matrix/genericextendablesparsematrixcsc.jl with Tm = SparseMatrixHIPCOO: host CSC,
    host findindex routing, device buffer for the misses; flush! = xmatrix + cscmatrix (:31-37)."""

    def __init__(self, m, n, Tm=SparseMatrixHIPCOO, **kw):
        self.Tm, self._kw = Tm, kw
        self.cscmatrix = SparseMatrixCSC(m, n)
        self.xmatrix = Tm(m, n, **kw)

    @property
    def shape(self):
        return self.cscmatrix.shape

    def nnznew(self):  # :21
        return self.xmatrix.nnz()

    def reset(self):  # :23-28
        m, n = self.cscmatrix.shape
        self.cscmatrix = SparseMatrixCSC(m, n)
        self.xmatrix = self.Tm(m, n, **self._kw)
        return self

    def flush(self):  # :31-37
        if self.xmatrix.nnz() > 0:
            self.cscmatrix = self.xmatrix + self.cscmatrix
            # :34 `ext.xmatrix = Tm(m,n)`: the flushed buffer IS an empty T_ext(m,n) again -- it is kept (its device
            # memory serves the next assembly; the Julia shim, which cannot change the wrapper, releases the dropped
            # buffer's memory eagerly instead: esp_release_buffers)
        return self

    def sparse(self):  # :39-42
        self.flush()
        return self.cscmatrix

    def nnz(self):
        return self.sparse().nnz()

    def arrays(self):
        return self.sparse().arrays()

    def __setitem__(self, ij, v):  # :44-54
        k = self.cscmatrix.findindex(*ij)
        if k > 0:
            self.cscmatrix.nzval[k - 1] = v
        else:
            self.xmatrix[ij] = v

    def __getitem__(self, ij):  # :57-66
        k = self.cscmatrix.findindex(*ij)
        if k > 0:
            return float(self.cscmatrix.nzval[k - 1])
        return self.xmatrix[ij]   # getindex(ext.xmatrix,i,j): the buffer's own (device-side) lookup

    def rawupdateindex(self, op, v, i, j):  # :68-79
        k = self.cscmatrix.findindex(i, j)
        if k > 0:
            self.cscmatrix.nzval[k - 1] = _apply(op, self.cscmatrix.nzval[k - 1], v)
        else:
            self.xmatrix.rawupdateindex(op, v, i, j)

    def updateindex(self, op, v, i, j):  # :81-92
        k = self.cscmatrix.findindex(i, j)
        if k > 0:
            self.cscmatrix.nzval[k - 1] = _apply(op, self.cscmatrix.nzval[k - 1], v)
        else:
            self.xmatrix.updateindex(op, v, i, j)


def _apply(op, a, b):
    return a - b if _op(op) == ESP_OP_SUB else a + b


class GenericMTExtendableSparseMatrixCSC:
    """src/matrix/genericmtextendablesparsematrixcsc.jl with Tm = SparseMatrixHIPCOO: one device
    buffer per partition `tid`; flush! = Base.sum(xmatrices, cscmatrix) (:45-51)."""

    def __init__(self, n, m, p=1, Tm=SparseMatrixHIPCOO, **kw):
        self.Tm, self._kw = Tm, kw
        self.cscmatrix = SparseMatrixCSC(m, n)
        self.xmatrices = [Tm(m, n, **kw) for _ in range(p)]
        self.colparts = np.array([1, 2], np.int64)
        self.partnodes = np.array([1, n + 1], np.int64)

    @property
    def shape(self):
        return self.cscmatrix.shape

    def partitioning(self, colparts, partnodes):  # partitioning!: :24-28
        self.partnodes = np.asarray(partnodes, np.int64)
        self.colparts = np.asarray(colparts, np.int64)
        return self

    def reset(self, p=None):  # :31-42
        m, n = self.cscmatrix.shape
        p = len(self.xmatrices) if p is None else p
        self.cscmatrix = SparseMatrixCSC(m, n)
        self._home = None
        self.xmatrices = [self.Tm(m, n, **self._kw) for _ in range(p)]
        self.colparts = np.array([1, 2], np.int64)
        self.partnodes = np.array([1, n + 1], np.int64)
        return self

    def nnznew(self):  # :84
        return sum(x.nnz() for x in self.xmatrices)

    def flush(self):  # :45-51
        m, n = self.cscmatrix.shape
        if self.Tm is SparseMatrixHIPCOO:
            # (the device CSC stays attached to a handle of the wrapper's between flushes; esp_flush_sum hands the buffers
            # back empty -- they ARE fresh T_ext(m,n) again, their device memory serves the next assembly)
            if getattr(self, "_home", None) is None:
                self._home = SparseMatrixHIPCOO(m, n, **self._kw)
            self.cscmatrix = SparseMatrixHIPCOO.sum(self.xmatrices, self.cscmatrix, home=self._home)
            return self
        self.cscmatrix = self.Tm.sum(self.xmatrices, self.cscmatrix)
        self.xmatrices = [self.Tm(m, n, **self._kw) for _ in range(len(self.xmatrices))]
        return self

    def sparse(self):
        self.flush()
        return self.cscmatrix

    def arrays(self):
        return self.sparse().arrays()

    def __setitem__(self, ij, v):  # :59-69
        k = self.cscmatrix.findindex(*ij)
        if k > 0:
            self.cscmatrix.nzval[k - 1] = v
        else:
            raise RuntimeError("use rawupdateindex! for new entries into GenericMTExtendableSparseMatrixCSC")

    def __getitem__(self, ij):  # :71-82
        k = self.cscmatrix.findindex(*ij)
        if k > 0:
            return float(self.cscmatrix.nzval[k - 1])
        if self.nnznew() == 0:
            return 0.0
        raise RuntimeError("flush! GenericMTExtendableSparseMatrixCSC before using getindex")

    def rawupdateindex(self, op, v, i, j, tid=1):  # :87-99
        k = self.cscmatrix.findindex(i, j)
        if k > 0:
            self.cscmatrix.nzval[k - 1] = _apply(op, self.cscmatrix.nzval[k - 1], v)
        else:
            self.xmatrices[tid - 1].rawupdateindex(op, v, i, j)

    def updateindex(self, op, v, i, j, tid=1):  # :102-114
        k = self.cscmatrix.findindex(i, j)
        if k > 0:
            self.cscmatrix.nzval[k - 1] = _apply(op, self.cscmatrix.nzval[k - 1], v)
        else:
            self.xmatrices[tid - 1].updateindex(op, v, i, j)
