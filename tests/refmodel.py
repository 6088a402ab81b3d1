"""Independent pure-Python/NumPy model of the assembly semantics (second oracle).

Deliberately written from the *specification* of the reference
(extendable.jl:159-255, sparsematrixlnk.jl:178-253) with a dict instead of the
linked list, so that it shares no code path with oracle/esparse_oracle.c.
"""
import numpy as np

SET, UPDATE, RAWUPDATE, PLUSEQ = 0, 1, 2, 3
COO = 13  # a triplet of sparse(I,J,V,m,n,+): always creates, first value as it is (device kind 3)


class DictModel:
    def __init__(self, m, n):
        self.m, self.n = m, n
        self.csc = {}      # (j,i) -> value, entries already flushed
        self.pending = {}  # (j,i) -> value, entries in the extension
        self.rebuilds = 0

    def _chk(self, i, j):
        if not (1 <= i <= self.m and 1 <= j <= self.n):
            raise IndexError((i, j))

    def apply(self, kind, v, i, j):
        self._chk(i, j)
        v = float(v)
        key = (j, i)
        if kind == PLUSEQ:  # A[i,j] += v : getindex then setindex!
            old = self.csc.get(key, self.pending.get(key, 0.0))
            kind, v = SET, old + v
        if key in self.csc:
            self.csc[key] = v if kind == SET else self.csc[key] + v
        elif key in self.pending:
            self.pending[key] = v if kind == SET else self.pending[key] + v
        elif kind == SET:
            if v != 0.0:
                self.pending[key] = v
        elif kind == UPDATE:
            if v != 0.0:
                self.pending[key] = 0.0 + v
        elif kind == COO:
            self.pending[key] = v
        else:
            self.pending[key] = 0.0 + v

    def flush(self):
        if self.pending:
            self.csc.update(self.pending)
            self.pending = {}
            self.rebuilds += 1

    def arrays(self):
        self.flush()
        keys = sorted(self.csc)
        colptr = np.ones(self.n + 1, np.int64)
        for (j, _i) in keys:
            colptr[j] += 1
        colptr = np.concatenate([[1], 1 + np.cumsum(colptr[1:] - 1)]).astype(np.int64)
        rowval = np.array([i for (_j, i) in keys], np.int64)
        nzval = np.array([self.csc[k] for k in keys], np.float64)
        return colptr, rowval, nzval

    def dropzeros(self):
        self.flush()
        self.csc = {k: v for k, v in self.csc.items() if v != 0.0}


def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def assert_csc_equal(a, b, what=""):
    (cp1, rv1, nz1), (cp2, rv2, nz2) = a, b
    assert np.array_equal(cp1, cp2), what + " colptr differs"
    assert np.array_equal(rv1, rv2), what + " rowval differs"
    assert np.array_equal(bits(nz1), bits(nz2)), what + " nzval differs (bitwise)"


def check_julia_invariants(m, n, colptr, rowval, nzval):
    """SparseMatrixCSC invariants (last bullet)."""
    assert colptr.shape == (n + 1,) and colptr[0] == 1
    assert np.all(np.diff(colptr) >= 0)
    z = colptr[-1] - 1
    assert rowval.shape == (z,) and nzval.shape == (z,)
    if z:
        assert rowval.min() >= 1 and rowval.max() <= m
        d = np.diff(rowval)
        starts = colptr[1:-1] - 1
        starts = starts[(starts > 0) & (starts < z)]
        interior = np.ones(max(z - 1, 0), bool)
        interior[starts - 1] = False
        assert np.all(d[interior] > 0), "rows not strictly increasing inside a column"
