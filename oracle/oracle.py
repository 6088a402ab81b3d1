"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module (see oracle/esparse_oracle.h).  The class
and method names mirror the reference (ExtendableSparse.jl) so that tests read
like the reference's own tests.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libesparse_oracle.so")

OP_ADD, OP_SUB = 0, 1
KIND_SET, KIND_UPDATE, KIND_RAWUPDATE, KIND_PLUSEQ = 0, 1, 2, 3
ERR_BOUNDS = -1


def build(force=False):
    """Compile the C restatement (gcc, seconds)."""
    src = [os.path.join(_HERE, f) for f in ("esparse_oracle.c", "esparse_oracle.h")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in src)):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B", "libesparse_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _SO


_native = None       # (CDLL, flags) of the box-native build used by the baseline timings
_native_tried = False


def _make_var(name):
    out = subprocess.check_output(["make", "-C", _HERE, "-pn"], stderr=subprocess.DEVNULL, text=True)
    for line in out.splitlines():
        if line.startswith(name + " ") and "=" in line:
            return line.split("=", 1)[1].strip()
    return ""


def build_native():
    """-O3 -march=native build of the same source ON THIS BOX (oracle/_native/, never shipped): what bench.py's
    cpu_baseline leg times.  Returns (CDLL, flags); falls back to the portable library if the compile fails."""
    global _native, _native_tried
    if _native_tried:
        return _native
    _native_tried = True
    so = os.path.join(_HERE, "_native", "libesparse_oracle_native.so")
    try:
        subprocess.check_call(["make", "-C", _HERE, "-B", "_native/libesparse_oracle_native.so"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        L = C.CDLL(so)
        flags = "gcc -O3 -march=native -ffp-contract=off, built on this box"
    except Exception:
        L = lib()
        flags = "gcc -O3 -ffp-contract=off (portable build; the -march=native build failed on this box)"
    for name in ("orc_bench_fdrand", "orc_bench_fdrand_mt"):
        f = getattr(L, name)
        f.restype = C.c_int64
    L.orc_bench_fdrand.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.orc_bench_fdrand_mt.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    _native = (L, flags)
    return _native


def build_flags():
    return build_native()[1]


_lib = None
i64 = C.c_int64
p_i64 = C.POINTER(C.c_int64)
p_f64 = C.POINTER(C.c_double)
p_u8 = C.POINTER(C.c_uint8)
vp = C.c_void_p


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("orc_csc_new", vp, i64, i64)
    sig("orc_csc_from_arrays", vp, i64, i64, p_i64, p_i64, p_f64)
    sig("orc_csc_free", None, vp)
    sig("orc_csc_m", i64, vp)
    sig("orc_csc_n", i64, vp)
    sig("orc_csc_nnz", i64, vp)
    sig("orc_csc_copy_out", None, vp, p_i64, p_i64, p_f64)
    sig("orc_csc_findindex", i64, vp, i64, i64)
    sig("orc_csc_pattern_equal", C.c_int, vp, vp)
    sig("orc_csc_pattern_hash", C.c_uint64, vp)
    sig("orc_csc_dropzeros", i64, vp)
    sig("orc_csc_jacobi", None, vp, p_f64)
    sig("orc_csc_ilu0", i64, vp, p_f64, p_i64)
    sig("orc_lnk_new", vp, i64, i64)
    sig("orc_lnk_from_csc", vp, vp)
    sig("orc_lnk_free", None, vp)
    sig("orc_lnk_nnz", i64, vp)
    sig("orc_lnk_nentries", i64, vp)
    sig("orc_lnk_setindex", C.c_int, vp, C.c_double, i64, i64)
    sig("orc_lnk_updateindex", C.c_int, vp, C.c_int, C.c_double, i64, i64)
    sig("orc_lnk_rawupdateindex", C.c_int, vp, C.c_int, C.c_double, i64, i64)
    sig("orc_lnk_getindex", C.c_int, vp, i64, i64, p_f64)
    sig("orc_lnk_plus_csc", vp, vp, vp)
    sig("orc_ext_new", vp, i64, i64)
    sig("orc_ext_from_csc", vp, vp)
    sig("orc_ext_free", None, vp)
    sig("orc_ext_setindex", C.c_int, vp, C.c_double, i64, i64)
    sig("orc_ext_updateindex", C.c_int, vp, C.c_int, C.c_double, i64, i64)
    sig("orc_ext_rawupdateindex", C.c_int, vp, C.c_int, C.c_double, i64, i64)
    sig("orc_ext_getindex", C.c_int, vp, i64, i64, p_f64)
    sig("orc_ext_flush", C.c_int, vp)
    sig("orc_ext_csc", vp, vp)
    sig("orc_ext_nnz", i64, vp)
    sig("orc_ext_pending", i64, vp)
    sig("orc_ext_phash", C.c_uint64, vp)
    sig("orc_ext_flush_count", i64, vp)
    sig("orc_ext_reset", None, vp)
    sig("orc_ext_zero_values", None, vp)
    sig("orc_ext_dropzeros", i64, vp)
    sig("orc_ext_apply", C.c_int, vp, i64, p_u8, p_i64, p_i64, p_f64, p_i64)
    sig("orc_sparse_coo", vp, i64, i64, i64, p_i64, p_i64, p_f64)
    sig("orc_csc_mul", None, vp, p_f64, p_f64)
    sig("orc_csc_mark_dirichlet", None, vp, C.c_double, p_u8)
    sig("orc_csc_eliminate_dirichlet", None, vp, p_u8)
    sig("orc_mt_new", vp, i64, i64, i64)
    sig("orc_mt_free", None, vp)
    sig("orc_mt_setindex", C.c_int, vp, C.c_double, i64, i64)
    sig("orc_mt_updateindex", C.c_int, vp, C.c_int, C.c_double, i64, i64, i64)
    sig("orc_mt_rawupdateindex", C.c_int, vp, C.c_int, C.c_double, i64, i64, i64)
    sig("orc_mt_apply", C.c_int, vp, i64, p_u8, p_i64, p_i64, p_f64, i64)
    sig("orc_mt_getindex", C.c_int, vp, i64, i64, p_f64)
    sig("orc_mt_nnznew", i64, vp)
    sig("orc_mt_flush", C.c_int, vp)
    sig("orc_mt_csc", vp, vp)
    sig("orc_mt_reset", None, vp)
    sig("orc_uniform", C.c_double, C.c_uint64, C.c_uint64)
    sig("orc_fdrand_count", i64, i64, i64, i64)
    sig("orc_fdrand_nnz", i64, i64, i64, i64)
    sig("orc_fdrand_stream", None, i64, i64, i64, C.c_int, C.c_uint64, p_i64, p_i64, p_f64)
    sig("orc_fdrand_ext", C.c_int, vp, i64, i64, i64, C.c_int, C.c_uint64, C.c_int)
    sig("orc_bench_fdrand", i64, i64, i64, i64, C.c_int, p_f64, p_f64)
    sig("orc_bench_fdrand_mt", i64, i64, i64, i64, i64, p_f64, p_f64)
    sig("orc_fem_ncells", i64, C.c_int, i64)
    sig("orc_fem_nnodes", i64, C.c_int, i64)
    sig("orc_fem_count", i64, C.c_int, i64)
    sig("orc_fem_cell_at", C.c_uint64, i64, i64, C.c_uint64, C.c_int)
    sig("orc_fem_cell_nodes", None, C.c_int, i64, i64, p_i64)
    sig("orc_fem_stream", None, C.c_int, i64, C.c_uint64, C.c_int, p_i64, p_i64, p_f64)
    sig("orc_fem_stream_range", None, C.c_int, i64, C.c_uint64, C.c_int, i64, i64, p_i64, p_i64, p_f64)
    sig("orc_fem_mesh_range", None, C.c_int, i64, C.c_uint64, C.c_int, C.c_int, C.c_uint64, i64, i64, p_i64, p_f64, p_f64)
    sig("orc_elements_stream", i64, C.c_int, i64, p_i64, p_f64, p_f64, p_i64, p_i64, p_f64)
    _lib = L
    return L


def _pi(a):
    return a.ctypes.data_as(p_i64)


def _pf(a):
    return a.ctypes.data_as(p_f64)


class BoundsError(IndexError):
    pass


def _check(rc):
    if rc == ERR_BOUNDS:
        raise BoundsError()
    if rc:
        raise RuntimeError("oracle error %d" % rc)


class CSC:
    """SparseMatrixCSC{Float64,Int64}; arrays hold 1-based values."""

    def __init__(self, m, n, colptr=None, rowval=None, nzval=None, _h=None, _own=True):
        L = lib()
        self._own = _own
        if _h is not None:
            self._h = _h
        elif colptr is None:
            self._h = L.orc_csc_new(m, n)
        else:
            cp = np.ascontiguousarray(colptr, dtype=np.int64)
            rv = np.ascontiguousarray(rowval, dtype=np.int64)
            nz = np.ascontiguousarray(nzval, dtype=np.float64)
            self._h = L.orc_csc_from_arrays(m, n, _pi(cp), _pi(rv), _pf(nz))

    def __del__(self):
        if getattr(self, "_own", False) and self._h:
            lib().orc_csc_free(self._h)
            self._h = None

    @property
    def shape(self):
        return (lib().orc_csc_m(self._h), lib().orc_csc_n(self._h))

    def nnz(self):
        return lib().orc_csc_nnz(self._h)

    def arrays(self):
        n = lib().orc_csc_n(self._h)
        z = self.nnz()
        cp = np.empty(n + 1, np.int64)
        rv = np.empty(z, np.int64)
        nz = np.empty(z, np.float64)
        lib().orc_csc_copy_out(self._h, _pi(cp), _pi(rv), _pf(nz))
        return cp, rv, nz

    def findindex(self, i, j):
        k = lib().orc_csc_findindex(self._h, i, j)
        _check(k if k < 0 else 0)
        return k

    def pattern_hash(self):
        return lib().orc_csc_pattern_hash(self._h)

    def pattern_equal(self, other):
        return bool(lib().orc_csc_pattern_equal(self._h, other._h))

    def dropzeros(self):
        return lib().orc_csc_dropzeros(self._h)

    def mark_dirichlet(self, penalty=1.0e20):
        out = np.zeros(self.shape[1], np.uint8)
        lib().orc_csc_mark_dirichlet(self._h, float(penalty), out.ctypes.data_as(p_u8))
        return out.astype(bool)

    def eliminate_dirichlet(self, marker):  # eliminate_dirichlet!(A, marker): in place
        mk = np.ascontiguousarray(np.asarray(marker) != 0, np.uint8)
        lib().orc_csc_eliminate_dirichlet(self._h, mk.ctypes.data_as(p_u8))
        return self

    def mul(self, x):  # mul!(r, A, x)
        x = np.ascontiguousarray(x, np.float64)
        m, n = self.shape
        assert len(x) == n
        r = np.empty(m, np.float64)
        lib().orc_csc_mul(self._h, _pf(x), _pf(r))
        return r

    def jacobi(self):
        out = np.empty(self.shape[1], np.float64)
        lib().orc_csc_jacobi(self._h, _pf(out))
        return out

    def ilu0(self):
        n = self.shape[1]
        xd, idg = np.empty(n, np.float64), np.empty(n, np.int64)
        bad = lib().orc_csc_ilu0(self._h, _pf(xd), _pi(idg))
        if bad:
            raise ValueError("column %d has no diagonal entry" % bad)
        return xd, idg

    def __add__(self, lnk):  # csc + lnk  (sparsematrixlnk.jl:385)
        return lnk + self


class SparseMatrixLNK:
    def __init__(self, m, n=None, _h=None):
        L = lib()
        if _h is not None:
            self._h = _h
        elif isinstance(m, CSC):
            self._h = L.orc_lnk_from_csc(m._h)
        else:
            self._h = L.orc_lnk_new(m, n)

    def __del__(self):
        if self._h:
            lib().orc_lnk_free(self._h)
            self._h = None

    def nnz(self):
        return lib().orc_lnk_nnz(self._h)

    def __setitem__(self, ij, v):
        _check(lib().orc_lnk_setindex(self._h, float(v), ij[0], ij[1]))

    def __getitem__(self, ij):
        out = C.c_double()
        _check(lib().orc_lnk_getindex(self._h, ij[0], ij[1], C.byref(out)))
        return out.value

    def updateindex(self, op, v, i, j):
        _check(lib().orc_lnk_updateindex(self._h, op, float(v), i, j))

    def rawupdateindex(self, op, v, i, j):
        _check(lib().orc_lnk_rawupdateindex(self._h, op, float(v), i, j))

    def __add__(self, csc):
        h = lib().orc_lnk_plus_csc(self._h, csc._h)
        if not h:
            raise ValueError("size mismatch")
        return CSC(0, 0, _h=h)


class ExtendableSparseMatrix:
    """ExtendableSparseMatrixCSC{Float64,Int64} (extendable.jl)."""

    def __init__(self, m, n=None):
        L = lib()
        if isinstance(m, CSC):
            self._h = L.orc_ext_from_csc(m._h)
        else:
            self._h = L.orc_ext_new(m, n)

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:   # (module globals may be gone at interpreter shutdown)
            lib().orc_ext_free(self._h)
            self._h = None

    def __setitem__(self, ij, v):
        _check(lib().orc_ext_setindex(self._h, float(v), ij[0], ij[1]))

    def __getitem__(self, ij):
        out = C.c_double()
        _check(lib().orc_ext_getindex(self._h, ij[0], ij[1], C.byref(out)))
        return out.value

    def updateindex(self, op, v, i, j):
        _check(lib().orc_ext_updateindex(self._h, op, float(v), i, j))

    def rawupdateindex(self, op, v, i, j):
        _check(lib().orc_ext_rawupdateindex(self._h, op, float(v), i, j))

    def apply(self, kinds, I, J, V):
        I = np.ascontiguousarray(I, np.int64)
        J = np.ascontiguousarray(J, np.int64)
        V = np.ascontiguousarray(V, np.float64)
        kp = None
        if kinds is not None:
            kinds = np.ascontiguousarray(kinds, np.uint8)
            kp = kinds.ctypes.data_as(p_u8)
        pos = C.c_int64(-1)
        rc = lib().orc_ext_apply(self._h, len(I), kp, _pi(I), _pi(J), _pf(V), C.byref(pos))
        _check(rc)

    def flush(self):
        return bool(lib().orc_ext_flush(self._h))

    def sparse(self):
        h = lib().orc_ext_csc(self._h)
        return CSC(0, 0, _h=h, _own=False)

    def arrays(self):
        return self.sparse().arrays()

    def nnz(self):
        return lib().orc_ext_nnz(self._h)

    def pending(self):
        return lib().orc_ext_pending(self._h)

    @property
    def phash(self):
        return lib().orc_ext_phash(self._h)

    def flush_count(self):
        return lib().orc_ext_flush_count(self._h)

    def reset(self):
        lib().orc_ext_reset(self._h)

    def zero_values(self):
        lib().orc_ext_zero_values(self._h)

    def dropzeros(self):
        return lib().orc_ext_dropzeros(self._h)

    def fdrand(self, nx, ny=1, nz=1, rand_mode=1, seed=0x5EED0002, style=KIND_PLUSEQ):
        rc = lib().orc_fdrand_ext(self._h, nx, ny, nz, rand_mode, seed, style)
        if rc:
            raise ValueError("Matrix size mismatch")
        return self


class MTExtendableSparseMatrix:
    """GenericMTExtendableSparseMatrixCSC{SparseMatrixDILNKC} (genericmt...jl)."""

    def __init__(self, m, n, nparts=1):
        self._h = lib().orc_mt_new(m, n, nparts)

    def __del__(self):
        if self._h:
            lib().orc_mt_free(self._h)
            self._h = None

    def __setitem__(self, ij, v):
        rc = lib().orc_mt_setindex(self._h, float(v), ij[0], ij[1])
        if rc == -2:
            raise RuntimeError("use rawupdateindex! for new entries")
        _check(rc)

    def __getitem__(self, ij):
        out = C.c_double()
        rc = lib().orc_mt_getindex(self._h, ij[0], ij[1], C.byref(out))
        if rc == -3:
            raise RuntimeError("flush! before using getindex")
        _check(rc)
        return out.value

    def updateindex(self, op, v, i, j, tid=1):
        _check(lib().orc_mt_updateindex(self._h, op, float(v), i, j, tid))

    def rawupdateindex(self, op, v, i, j, tid=1):
        _check(lib().orc_mt_rawupdateindex(self._h, op, float(v), i, j, tid))

    def apply(self, kinds, I, J, V, tid=1):
        """a batch of per-entry calls with one tid, in order (kinds None: rawupdateindex!)"""
        I = np.ascontiguousarray(I, np.int64)
        J = np.ascontiguousarray(J, np.int64)
        V = np.ascontiguousarray(V, np.float64)
        kk = None if kinds is None else np.ascontiguousarray(kinds, np.uint8)
        kp = None if kk is None else kk.ctypes.data_as(C.POINTER(C.c_uint8))
        rc = lib().orc_mt_apply(self._h, len(I), kp, _pi(I), _pi(J), _pf(V), tid)
        if rc == -2:
            raise RuntimeError("use rawupdateindex! for new entries")
        _check(rc)

    def nnznew(self):
        return lib().orc_mt_nnznew(self._h)

    def flush(self):
        return bool(lib().orc_mt_flush(self._h))

    def sparse(self):
        return CSC(0, 0, _h=lib().orc_mt_csc(self._h), _own=False)

    def arrays(self):
        return self.sparse().arrays()

    def reset(self):
        lib().orc_mt_reset(self._h)


def sparse_coo(I, J, V, m=None, n=None):
    """sparse(I,J,V[,m,n]) with combine = + : what the COO constructors of extendable.jl:92-104 build."""
    I = np.ascontiguousarray(I, np.int64)
    J = np.ascontiguousarray(J, np.int64)
    V = np.ascontiguousarray(V, np.float64)
    m = int(I.max()) if m is None else m
    n = int(J.max()) if n is None else n
    h = lib().orc_sparse_coo(m, n, len(I), _pi(I), _pi(J), _pf(V))
    if not h:
        raise BoundsError()
    return CSC(0, 0, _h=h)


def uniform(seed, counter):
    return lib().orc_uniform(seed, counter)


def fdrand_count(nx, ny=1, nz=1):
    return lib().orc_fdrand_count(nx, ny, nz)


def fdrand_nnz(nx, ny=1, nz=1):
    return lib().orc_fdrand_nnz(nx, ny, nz)


def fdrand_stream(nx, ny=1, nz=1, rand_mode=1, seed=0x5EED0002):
    e = fdrand_count(nx, ny, nz)
    I = np.empty(e, np.int64)
    J = np.empty(e, np.int64)
    V = np.empty(e, np.float64)
    lib().orc_fdrand_stream(nx, ny, nz, rand_mode, seed, _pi(I), _pi(J), _pf(V))
    return I, J, V


def fdrand(nx, ny=1, nz=1, rand_mode=1, seed=0x5EED0002, style=KIND_PLUSEQ):
    """fdrand(Float64,nx,ny,nz; matrixtype=ExtendableSparseMatrix) (sprand.jl:226-256)."""
    N = nx * ny * nz
    return ExtendableSparseMatrix(N, N).fdrand(nx, ny, nz, rand_mode, seed, style)


def bench_fdrand(nx, ny, nz, style=KIND_UPDATE):
    ti, tf = C.c_double(), C.c_double()
    z = build_native()[0].orc_bench_fdrand(nx, ny, nz, style, C.byref(ti), C.byref(tf))
    return z, ti.value, tf.value


def bench_fdrand_mt(nx, ny, nz, nthreads):
    """cpu_baseline.mt of bench.py: nthreads buffers filled in parallel + serial COO merge (see orc_bench_fdrand_mt)."""
    ti, tf = C.c_double(), C.c_double()
    z = build_native()[0].orc_bench_fdrand_mt(nx, ny, nz, nthreads, C.byref(ti), C.byref(tf))
    return z, ti.value, tf.value


def fem_sizes(dim, npd):
    L = lib()
    return L.orc_fem_nnodes(dim, npd), L.orc_fem_ncells(dim, npd), L.orc_fem_count(dim, npd)


def fem_cell_at(pos, ncells, seed, order_mode=1):
    return lib().orc_fem_cell_at(pos, ncells, seed, order_mode)


def fem_cell_nodes(dim, npd, cell):
    out = np.empty(dim + 1, np.int64)
    lib().orc_fem_cell_nodes(dim, npd, cell, _pi(out))
    return out


def fem_stream_range(dim, npd, p0, p1, seed=0x5EED0004, order_mode=1):
    """The updates of the cells at stream positions [p0, p1) (bench-size digests feed the oracle in chunks)."""
    e = (p1 - p0) * (dim + 1) * (dim + 2)
    I = np.empty(e, np.int64)
    J = np.empty(e, np.int64)
    V = np.empty(e, np.float64)
    lib().orc_fem_stream_range(dim, npd, seed, order_mode, p0, p1, _pi(I), _pi(J), _pf(V))
    return I, J, V


def fem_stream(dim, npd, seed=0x5EED0004, order_mode=1):
    e = lib().orc_fem_count(dim, npd)
    I = np.empty(e, np.int64)
    J = np.empty(e, np.int64)
    V = np.empty(e, np.float64)
    lib().orc_fem_stream(dim, npd, seed, order_mode, _pi(I), _pi(J), _pf(V))
    return I, J, V


def fem_mesh(dim, npd, seed=0x5EED0004, order_mode=1, node_mode=0, node_seed=0x5EED0014, p0=0, p1=None, diag=True):
    """Element data of the cells at stream positions [p0, p1) as a caller of testassemble! holds it: cellnodes
    (nloc x nc, Fortran order like Julia's grid[CellNodes]), elmat = vol * S (nloc x nloc x nc), diag = 0.1 * vol / (dim+1)."""
    L = lib()
    nloc = dim + 1
    if p1 is None:
        p1 = L.orc_fem_ncells(dim, npd)
    nc = p1 - p0
    cn = np.empty((nloc, nc), np.int64, order="F")
    em = np.empty((nloc, nloc, nc), np.float64, order="F")
    dg = np.empty((nloc, nc), np.float64, order="F") if diag else None
    L.orc_fem_mesh_range(dim, npd, seed, order_mode, node_mode, node_seed, p0, p1, _pi(cn), _pf(em),
                         _pf(dg) if diag else None)
    return cn, em, dg


def elements_stream(cellnodes, elmat, diag=None):
    """The update calls of test/femtools.jl:62-69 for element data held in arrays (Julia layouts), as triplets."""
    nloc, nc = cellnodes.shape
    cn = np.asfortranarray(cellnodes, np.int64)
    em = np.asfortranarray(elmat, np.float64)
    assert em.shape == (nloc, nloc, nc)
    dg = None if diag is None else np.asfortranarray(diag, np.float64)
    e = nc * nloc * (nloc + (0 if diag is None else 1))
    I = np.empty(e, np.int64)
    J = np.empty(e, np.int64)
    V = np.empty(e, np.float64)
    got = lib().orc_elements_stream(nloc, nc, _pi(cn), _pf(em), _pf(dg) if dg is not None else None, _pi(I), _pi(J), _pf(V))
    assert got == e
    return I, J, V
