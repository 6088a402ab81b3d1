"""Pins the CPU oracle (oracle/esparse_oracle.c).

The reference stores no golden vectors and cannot run
here (Julia absent), so the oracle is pinned against
  * the known-answer tests of the reference's own suite for this path
    (cited per test), and
  * an independent dict/NumPy/SciPy restatement (tests/refmodel.py).
"""
import numpy as np
import pytest
import scipy.sparse as sp

from refmodel import (PLUSEQ, RAWUPDATE, SET, UPDATE, DictModel, assert_csc_equal, bits,
                      check_julia_invariants)


# ---------------------------------------------------------------- test_updates.jl
def test_updates_nnz_trace(orc):
    """test/test_updates.jl:10-25 -- nnz 0,2,2,3,(dropzeros)2,3,3."""
    A = orc.ExtendableSparseMatrix(10, 10)
    assert A.nnz() == 0
    A[1, 3] = 5
    A.updateindex(orc.OP_ADD, 6.0, 4, 5)
    A.updateindex(orc.OP_ADD, 0.0, 2, 3)
    assert A.nnz() == 2
    A.rawupdateindex(orc.OP_ADD, 0.0, 2, 3)
    assert A.nnz() == 3
    A.dropzeros()
    assert A.nnz() == 2
    A.rawupdateindex(orc.OP_ADD, 0.1, 2, 3)
    assert A.nnz() == 3
    A.dropzeros()
    assert A.nnz() == 3


# --------------------------------------------------------------- test_assembly.jl
def _assembly(orc, m, n, xnnz, nsplice, seed):
    """test/test_assembly.jl:6-35: S[i,j]+=a vs A[i,j]+=a, `==` after every flush."""
    rng = np.random.default_rng(seed)
    A = orc.ExtendableSparseMatrix(m, n)
    S = {}
    model = DictModel(m, n)
    for _ in range(nsplice):
        I = rng.integers(1, m + 1, xnnz)
        J = rng.integers(1, n + 1, xnnz)
        a = 1.0 + rng.random(xnnz)
        for i, j, v in zip(I, J, a):
            S[(j, i)] = S.get((j, i), 0.0) + v
            model.apply(PLUSEQ, v, int(i), int(j))
        A.apply(np.full(xnnz, PLUSEQ, np.uint8), I, J, a)
        A.flush()
        cp, rv, nz = A.arrays()
        check_julia_invariants(m, n, cp, rv, nz)          # :19-21 rows sorted
        assert len(rv) == len(S)                          # :22
        col = np.repeat(np.arange(1, n + 1), np.diff(cp))
        for j, i, v in zip(col, rv, nz):                  # :24-32 `==`, both directions
            assert S[(int(j), int(i))] == v
        assert_csc_equal((cp, rv, nz), model.arrays())


@pytest.mark.parametrize("m,n,xnnz,nsplice", [
    (10, 10, 5, 1), (100, 100, 500, 2), (1000, 1000, 5000, 3),      # :37-39
    (20, 10, 5, 1), (200, 100, 500, 2), (2000, 1000, 5000, 3),      # :41-43
    (10, 20, 5, 1), (100, 200, 500, 2), (1000, 2000, 5000, 3),      # :45-47
])
def test_assembly_fixed_shapes(orc, m, n, xnnz, nsplice):
    _assembly(orc, m, n, xnnz, nsplice, seed=m * 7 + n)


def test_assembly_random_shapes(orc):
    """test/test_assembly.jl:49-55."""
    rng = np.random.default_rng(1234)
    for _ in range(10):
        m, n, z = (int(rng.integers(1, 10001)) for _ in range(3))
        _assembly(orc, m, n, z, int(rng.integers(1, 6)), seed=z)


def test_mixed_kinds_against_dict_model(orc):
    """SET/UPDATE/RAWUPDATE mixes incl. zeros and -0.0, several flushes."""
    rng = np.random.default_rng(7)
    m, n = 37, 23
    A = orc.ExtendableSparseMatrix(m, n)
    M = DictModel(m, n)
    pool = np.array([0.0, -0.0, 1.5, -1.5, 1e-300, 3.25, -7.0])
    for splice in range(4):
        cnt = 600
        kinds = rng.integers(0, 4, cnt).astype(np.uint8)
        I = rng.integers(1, m + 1, cnt)
        J = rng.integers(1, n + 1, cnt)
        V = np.where(rng.random(cnt) < 0.4, rng.choice(pool, cnt), rng.standard_normal(cnt))
        A.apply(kinds, I, J, V)
        for k, i, j, v in zip(kinds, I, J, V):
            M.apply(int(k), v, int(i), int(j))
        if splice % 2 == 0:
            A.flush()
            M.flush()
    assert_csc_equal(A.arrays(), M.arrays())


def test_bounds_error(orc):
    """sparsematrixcsc.jl:8-10: out-of-range throws before anything is appended."""
    A = orc.ExtendableSparseMatrix(5, 4)
    for (i, j) in [(0, 1), (6, 1), (1, 0), (1, 5)]:
        with pytest.raises(IndexError):
            A.updateindex(orc.OP_ADD, 1.0, i, j)
        with pytest.raises(IndexError):
            A.rawupdateindex(orc.OP_ADD, 1.0, i, j)
        with pytest.raises(IndexError):
            A[i, j] = 1.0
    assert A.pending() == 0 and A.nnz() == 0


def test_flush_gate_and_phash(orc):
    """extendable.jl:248-255 + SURVEY appendix A: zero-only buffer is not flushed."""
    A = orc.ExtendableSparseMatrix(6, 6)
    assert A.phash == 0
    A.updateindex(orc.OP_ADD, 0.0, 2, 2)
    A[3, 3] = 0.0
    assert not A.flush() and A.phash == 0 and A.flush_count() == 0
    A.rawupdateindex(orc.OP_ADD, 0.0, 2, 2)
    assert A.flush() and A.phash != 0
    h = A.phash
    A.updateindex(orc.OP_ADD, 2.0, 2, 2)       # hit in CSC: in place, no rebuild
    assert not A.flush() and A.phash == h
    assert A[2, 2] == 2.0
    A.reset()                                   # extendable.jl:269-272: phash kept
    assert A.nnz() == 0 and A.phash == h


# ----------------------------------------------------------------- test_fdrand.jl
def _analytic_fd(nx, ny, nz):
    """The rand=()->1 matrix (sprand.jl:94-120) built independently with SciPy."""
    N = nx * ny * nz
    hx, hy, hz = 1.0 / nx, 1.0 / ny, 1.0 / nz
    D = {}

    def add(i, j, v):
        D[(i, j)] = D.get((i, j), 0.0) + v

    l = 1
    for k in range(1, nz + 1):
        for j in range(1, ny + 1):
            for i in range(1, nx + 1):
                if i < nx:
                    w = 1.0 * hy * hz / hx
                    add(l, l + 1, -w); add(l + 1, l, -w); add(l, l, w); add(l + 1, l + 1, w)
                if i == 1 or i == nx:
                    add(l, l, 1.0 * hy * hz)
                if j < ny:
                    w = 1.0 * hx * hz / hy
                    add(l, l + nx, -w); add(l + nx, l, -w); add(l, l, w); add(l + nx, l + nx, w)
                if ny > 2 and (j == 1 or j == ny):
                    add(l, l, 1.0 * hx * hz)
                if k < nz:
                    w = 1.0 * hx * hy / hz
                    nxy = nx * ny
                    add(l, l + nxy, -w); add(l + nxy, l, -w); add(l, l, w); add(l + nxy, l + nxy, w)
                if nz > 2 and (k == 1 or k == nz):
                    add(l, l, 1.0 * hx * hy)
                l += 1
    ij = np.array(sorted(D, key=lambda t: (t[1], t[0])))
    v = np.array([D[tuple(t)] for t in ij])
    return sp.csc_matrix((v, (ij[:, 0] - 1, ij[:, 1] - 1)), shape=(N, N))


def _to_scipy(arrs, m, n):
    cp, rv, nz = arrs
    return sp.csc_matrix((nz, rv - 1, cp - 1), shape=(m, n))


@pytest.mark.parametrize("dims", [(100, 1, 1), (10, 10, 1), (5, 5, 5)])
def test_fdrand_update_styles_and_coo(orc, dims):
    """test/test_fdrand.jl:22-53: `+=` == rawupdateindex! == updateindex! == COO route."""
    nx, ny, nz = dims
    N = nx * ny * nz
    A1 = orc.fdrand(nx, ny, nz, rand_mode=0, style=orc.KIND_PLUSEQ).arrays()
    A2 = orc.fdrand(nx, ny, nz, rand_mode=0, style=orc.KIND_RAWUPDATE).arrays()
    A3 = orc.fdrand(nx, ny, nz, rand_mode=0, style=orc.KIND_UPDATE).arrays()
    assert_csc_equal(A1, A2)
    assert_csc_equal(A1, A3)
    check_julia_invariants(N, N, *A1)
    I, J, V = orc.fdrand_stream(nx, ny, nz, rand_mode=0)
    coo = sp.coo_matrix((V, (I - 1, J - 1)), shape=(N, N)).tocsc()
    coo.sort_indices()
    mine = _to_scipy(A1, N, N)
    assert np.array_equal(coo.indptr, mine.indptr) and np.array_equal(coo.indices, mine.indices)
    assert np.allclose(coo.data, mine.data, rtol=1e-13, atol=0)
    ana = _analytic_fd(nx, ny, nz)
    assert np.array_equal(ana.indptr, mine.indptr) and np.array_equal(ana.indices, mine.indices)
    assert np.array_equal(bits(ana.data), bits(mine.data))
    assert len(A1[1]) == orc.fdrand_nnz(nx, ny, nz)
    assert len(I) == orc.fdrand_count(nx, ny, nz)


@pytest.mark.parametrize("dims", [(100, 1, 1), (10, 10, 1), (5, 5, 5)])
def test_fdrand_m_matrix(orc, dims):
    """test/test_fdrand.jl:13-19: Jacobi spectral radius < 1, inverse > 0."""
    nx, ny, nz = dims
    N = nx * ny * nz
    A = _to_scipy(orc.fdrand(nx, ny, nz, rand_mode=1).arrays(), N, N).toarray()
    Jm = np.eye(N) - np.diag(1.0 / np.diag(A)) @ A
    ev = np.linalg.eigvals(Jm).real
    assert abs(ev.min()) < 1 and abs(ev.max()) < 1
    assert np.linalg.inv(A).min() > 0


def test_fdrand_counts_match_survey(orc):
    """SURVEY.md section 8 sizes."""
    assert orc.fdrand_count(30, 30, 30) == 318600 and orc.fdrand_nnz(30, 30, 30) == 183600
    assert orc.fdrand_count(256, 256, 256) == 200933376
    assert orc.fdrand_nnz(256, 256, 256) == 117047296
    assert orc.fdrand_count(512, 512, 512) == 1609039872
    assert orc.fdrand_nnz(512, 512, 512) == 937951232


def test_fdrand_reassembly_hits_csc(orc):
    """docs/src/example.md:203-219 / SURVEY 3.2: second fdrand! only hits the CSC."""
    A = orc.fdrand(6, 5, 4, rand_mode=1, seed=11)
    first = A.arrays()
    n0 = A.flush_count()
    A.fdrand(6, 5, 4, rand_mode=1, seed=11, style=orc.KIND_UPDATE)
    assert A.flush_count() == n0                      # no rebuild
    second = A.arrays()
    assert np.array_equal(first[0], second[0]) and np.array_equal(first[1], second[1])
    # zero! then the same in-order accumulation starting from 0.0 -> same bits
    assert np.array_equal(bits(first[2]), bits(second[2]))


# ------------------------------------------- test_operations.jl / test_constructors.jl
def _sprand_csc(orc, rng, m, n, d):
    S = sp.random(m, n, density=d, format="csc", random_state=rng, dtype=np.float64)
    S.sort_indices()
    return orc.CSC(m, n, S.indptr.astype(np.int64) + 1, S.indices.astype(np.int64) + 1, S.data), S


def test_csc_plus_lnk_is_2csc(orc):
    """test/test_operations.jl:8-13,25-30."""
    rng = np.random.default_rng(5)
    for _ in range(10):
        m, n = int(rng.integers(1, 1001)), int(rng.integers(1, 1001))
        csc, S = _sprand_csc(orc, rng, m, n, 0.3 * rng.random())
        lnk = orc.SparseMatrixLNK(csc)
        cp, rv, nz = (csc + lnk).arrays()
        assert np.array_equal(cp, S.indptr + 1) and np.array_equal(rv, S.indices + 1)
        assert np.array_equal(bits(nz), bits(2 * S.data))


def test_lnk_csc_round_trip(orc):
    """test/test_constructors.jl:26-31,59-64."""
    rng = np.random.default_rng(6)
    for _ in range(10):
        m, n = int(rng.integers(1, 1001)), int(rng.integers(1, 1001))
        csc, S = _sprand_csc(orc, rng, m, n, 0.3 * rng.random())
        lnk = orc.SparseMatrixLNK(csc)
        back = lnk + orc.CSC(m, n)
        assert_csc_equal(back.arrays(), (S.indptr + 1, S.indices + 1, S.data))


def test_coo_constructor_equals_scipy(orc):
    """test/test_constructors.jl:48-51 (sparse(I,J,A) == sparse(Ext(I,J,A))) via updateindex!."""
    rng = np.random.default_rng(8)
    I = rng.integers(1, 11, 100)
    J = rng.integers(1, 11, 100)
    V = rng.random(100)
    A = orc.ExtendableSparseMatrix(10, 10)
    A.apply(np.full(100, UPDATE, np.uint8), I, J, V)
    M = DictModel(10, 10)
    for i, j, v in zip(I, J, V):
        M.apply(UPDATE, v, int(i), int(j))
    assert_csc_equal(A.arrays(), M.arrays())
    S = sp.coo_matrix((V, (I - 1, J - 1)), shape=(10, 10)).tocsc()
    S.sort_indices()
    cp, rv, nz = A.arrays()
    assert np.array_equal(cp - 1, S.indptr) and np.array_equal(rv - 1, S.indices)
    assert np.allclose(nz, S.data, rtol=1e-14)


@pytest.mark.parametrize("dims", [(1000, 1, 1), (20, 20, 1), (10, 10, 10)])
def test_dirichlet_known_answer(orc, dims):
    """test/test_dirichlet.jl:8-20: penalty rows A[i,i]=1e30 vs mark_dirichlet + eliminate_dirichlet: both
    systems have the same solution (right-hand side zeroed at the marked nodes)."""
    import scipy.sparse.linalg as spla
    nx, ny, nz = dims
    N = nx * ny * nz
    A = orc.fdrand(nx, ny, nz, rand_mode=1, seed=3, style=orc.KIND_UPDATE)
    for i in range(1, N + 1, 10):
        A[i, i] = 1.0e30
    f = np.ones(N)
    u = spla.spsolve(_to_scipy(A.arrays(), N, N).tocsc(), f)
    C0 = A.sparse()
    diri = C0.mark_dirichlet()
    assert diri.sum() == len(range(1, N + 1, 10)) and diri[0] and not diri[1]
    fD = f * (1 - diri)
    C0.eliminate_dirichlet(diri)
    AD = _to_scipy(C0.arrays(), N, N).tocsc()
    uD = spla.spsolve(AD, fD)
    assert np.max(np.abs(uD - u)) <= 1e-9 * max(1.0, np.max(np.abs(u)))
    # structure of the eliminated matrix: unit rows/columns at the marked nodes, pattern unchanged
    D = AD.toarray() if N <= 1000 else None
    if D is not None:
        for i in np.flatnonzero(diri):
            assert D[i, i] == 1.0 and np.count_nonzero(D[i, :]) == 1 and np.count_nonzero(D[:, i]) == 1


def test_mul_restatement(orc):
    """mul!(r, A, x) column loop == SciPy's product (to rounding), r .= 0 first, empty matrix gives zeros."""
    rng = np.random.default_rng(19)
    m, n, cnt = 60, 45, 500
    I = rng.integers(1, m + 1, cnt)
    J = rng.integers(1, n + 1, cnt)
    V = rng.standard_normal(cnt)
    A = orc.sparse_coo(I, J, V, m, n)
    x = rng.standard_normal(n)
    S = sp.coo_matrix((V, (I - 1, J - 1)), shape=(m, n)).tocsr()
    assert np.allclose(A.mul(x), S @ x, rtol=1e-12, atol=1e-13)
    # the exact summation order: row i adds its entries in increasing column order, starting from 0.0
    cp, rv, nz = A.arrays()
    r = np.zeros(m)
    for col in range(n):
        for k in range(cp[col] - 1, cp[col + 1] - 1):
            r[rv[k] - 1] = r[rv[k] - 1] + nz[k] * x[col]
    assert np.array_equal(bits(r), bits(A.mul(x)))
    assert np.array_equal(orc.CSC(4, 3).mul(np.ones(3)), np.zeros(4))


def test_sparse_coo_restatement(orc):
    """sparse(I,J,V,m,n,+) of the COO constructors (extendable.jl:92-104, test_constructors.jl:48-51):
    structure = SciPy's, values = left-to-right sums in input order with the first value as it is
    (dict model), numerical zeros kept, -0.0 survives, bounds rejected."""
    from refmodel import COO
    rng = np.random.default_rng(18)
    m, n, cnt = 37, 29, 900
    I = rng.integers(1, m + 1, cnt)
    J = rng.integers(1, n + 1, cnt)
    V = np.where(rng.random(cnt) < 0.2, 0.0, rng.standard_normal(cnt))
    A = orc.sparse_coo(I, J, V, m, n)
    M = DictModel(m, n)
    for i, j, v in zip(I, J, V):
        M.apply(COO, v, int(i), int(j))
    assert_csc_equal(A.arrays(), M.arrays())
    check_julia_invariants(m, n, *A.arrays())
    S = sp.coo_matrix((V, (I - 1, J - 1)), shape=(m, n)).tocsc()
    S.sort_indices()
    cp, rv, nz = A.arrays()
    assert np.array_equal(cp - 1, S.indptr) and np.array_equal(rv - 1, S.indices)   # zeros are kept by both
    assert np.allclose(nz, S.data, rtol=1e-13, atol=1e-15)
    # sizes default to the largest indices; a lone -0.0 keeps its sign (no 0.0 + v)
    B = orc.sparse_coo([2, 3, 3], [4, 1, 1], [-0.0, 1.5, 2.0])
    assert B.shape == (3, 4)
    cp, rv, nz = B.arrays()
    assert list(cp) == [1, 2, 2, 2, 3] and list(rv) == [3, 2] and nz[0] == 3.5 and np.signbit(nz[1])
    with pytest.raises(orc.BoundsError):
        orc.sparse_coo([1, 5], [1, 1], [1.0, 2.0], 4, 4)
    # the stencil through the COO route equals the updateindex! route (test_fdrand.jl:47-53)
    I, J, V = orc.fdrand_stream(5, 4, 3, rand_mode=1, seed=4)
    assert_csc_equal(orc.sparse_coo(I, J, V, 60, 60).arrays(), orc.fdrand(5, 4, 3, rand_mode=1, seed=4, style=orc.KIND_UPDATE).arrays())


# ------------------------------------------------------------- MT wrapper (a16,a17)
def test_mt_wrapper_sum_semantics(orc):
    """genericmtextendablesparsematrixcsc.jl:45-114 + sparsematrixdilnkc.jl:397-435."""
    A = orc.MTExtendableSparseMatrix(8, 8, 3)
    with pytest.raises(RuntimeError):
        A[1, 1] = 2.0                       # :67 new entries must use rawupdateindex!
    A.rawupdateindex(orc.OP_ADD, 0.1, 2, 3, 1)
    A.rawupdateindex(orc.OP_ADD, 0.2, 2, 3, 2)
    A.rawupdateindex(orc.OP_ADD, 0.3, 2, 3, 3)
    A.rawupdateindex(orc.OP_ADD, 0.7, 2, 3, 1)
    A.updateindex(orc.OP_ADD, 0.0, 5, 5, 2)  # no entry
    with pytest.raises(RuntimeError):
        A[2, 3]                              # :80 flush before getindex
    assert A.nnznew() == 3
    A.flush()
    cp, rv, nz = A.arrays()
    assert len(rv) == 1 and rv[0] == 2 and cp[3] - cp[2] == 1
    assert nz[0] == ((0.0 + 0.1 + 0.7) + (0.0 + 0.2)) + (0.0 + 0.3)
    A[2, 3] = 9.0                            # existing entry: allowed (:64-65)
    A.rawupdateindex(orc.OP_ADD, 1.0, 2, 3, 2)
    assert A.nnznew() == 0 and A[2, 3] == 10.0


# ------------------------------------------------------------------- FEM stream (a19)
@pytest.mark.parametrize("dim,npd", [(2, 6), (3, 4)])
def test_fem_stream_properties(orc, dim, npd):
    """test/femtools.jl:45-72 pattern on the build's own Kuhn grid: row sums of the
    stiffness part vanish, mass part sums to 0.1*volume, matrix symmetric, and a
    permuted cell order gives the same matrix up to rounding."""
    nn, nc, cnt = orc.fem_sizes(dim, npd)
    assert cnt == nc * (dim + 1) * (dim + 2)
    perm = [orc.fem_cell_at(p, nc, 0x5EED0004, 1) for p in range(nc)]
    assert sorted(perm) == list(range(nc)) and perm != list(range(nc))
    mats = []
    for order in (0, 1):
        I, J, V = orc.fem_stream(dim, npd, order_mode=order)
        assert len(I) == cnt and I.min() >= 1 and I.max() <= nn
        A = orc.ExtendableSparseMatrix(nn, nn)
        A.apply(np.full(cnt, RAWUPDATE, np.uint8), I, J, V)
        mats.append(_to_scipy(A.arrays(), nn, nn))
    A0, A1 = mats
    assert np.array_equal(A0.indptr, A1.indptr) and np.array_equal(A0.indices, A1.indices)
    assert np.allclose(A0.data, A1.data, rtol=1e-12, atol=1e-15)
    assert abs(A0 - A0.T).max() < 1e-14
    assert np.isclose(A0.sum(), 0.1 * 1.0, rtol=1e-12)   # stiffness rows sum to 0; mass = 0.1*|Omega|
    # every cell has dim+1 distinct nodes inside the grid
    for c in range(0, nc, max(1, nc // 17)):
        nodes = orc.fem_cell_nodes(dim, npd, c)
        assert len(set(nodes.tolist())) == dim + 1


def test_elements_stream_is_the_fem_loop(orc):
    """orc_elements_stream restates the loops of test/femtools.jl:61-69 over arrays (cellnodes, elmat = vol * S, diag):
    fed the mesh arrays of the Kuhn grid it must give the very stream orc_fem_stream emits (same calls, same order, same
    bits); an independent NumPy restatement of the loop nest agrees for arbitrary arrays; a permuted node numbering is a
    relabelling of rows and columns."""
    for dim, npd in ((2, 9), (3, 5)):
        cn, em, dg = orc.fem_mesh(dim, npd, seed=0x5EED0004, order_mode=1)
        I, J, V = orc.elements_stream(cn, em, dg)
        I0, J0, V0 = orc.fem_stream(dim, npd, seed=0x5EED0004, order_mode=1)
        assert np.array_equal(I, I0) and np.array_equal(J, J0) and np.array_equal(V.view(np.uint64), V0.view(np.uint64))
        # chunks of cells concatenate
        nc = cn.shape[1]
        ca, ea, da = orc.fem_mesh(dim, npd, p0=0, p1=nc // 3)
        cb, eb, db = orc.fem_mesh(dim, npd, p0=nc // 3, p1=nc)
        assert np.array_equal(np.concatenate([ca, cb], axis=1), cn) and np.array_equal(np.concatenate([ea, eb], axis=2), em)
        # permuted numbering: a bijection of the nodes applied to the connectivity, element data unchanged
        cp, ep, dp = orc.fem_mesh(dim, npd, node_mode=1)
        nn = npd ** dim
        perm = np.zeros(nn + 1, np.int64)
        perm[cn.ravel(order="F")] = cp.ravel(order="F")
        assert sorted(perm[1:].tolist()) == list(range(1, nn + 1)) and not np.array_equal(perm[1:], np.arange(1, nn + 1))
        assert np.array_equal(perm[cn], cp) and np.array_equal(ep, em) and np.array_equal(dp, dg)
    rng = np.random.default_rng(3)
    for nloc, diag in ((1, True), (3, False), (5, True)):
        nc = 17
        cn = np.asfortranarray(rng.integers(1, 50, (nloc, nc)))
        em = np.asfortranarray(rng.standard_normal((nloc, nloc, nc)))
        dg = np.asfortranarray(rng.standard_normal((nloc, nc))) if diag else None
        I, J, V = orc.elements_stream(cn, em, dg)
        ref = []
        for c in range(nc):               # femtools.jl:61-69
            for il in range(nloc):
                i = cn[il, c]
                if diag:
                    ref.append((i, i, dg[il, c]))
                for jl in range(nloc):
                    ref.append((i, cn[jl, c], em[il, jl, c]))
        assert [(a, b, v) for a, b, v in zip(I.tolist(), J.tolist(), V.tolist())] == [(int(a), int(b), float(v)) for a, b, v in ref]


def test_uniform_is_splitmix64(orc):
    def mix(z):
        M = (1 << 64) - 1
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)
    for seed, ctr in [(0, 0), (0x5EED0002, 12345), (2**63 + 5, 2**40)]:
        z = mix((seed + (ctr + 1) * 0x9E3779B97F4A7C15) & ((1 << 64) - 1))
        assert orc.uniform(seed, ctr) == (z >> 11) * 2.0 ** -53


def test_jacobi_and_ilu0_setup_restatement(orc):
    """jacobi.jl:5-12 and ilu0.jl:8-41: the literal loops give invdiag = 1 ./ diag(A) and, for ILU0, idiag = position of
    the diagonal and xdiag = 1 ./ diag(A) as well (every `xdiag[i] -= ...` of an earlier column is overwritten)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(5)
    n = 40
    M = sp.random(n, n, density=0.15, random_state=3, format="csc") + sp.diags(2.0 + rng.random(n))
    M = sp.csc_matrix(M)
    M.sort_indices()
    C0 = orc.CSC(n, n, M.indptr + 1, M.indices + 1, M.data)
    d = M.diagonal()
    assert np.array_equal(C0.jacobi(), 1.0 / d)
    xd, idg = C0.ilu0()
    assert np.array_equal(xd, 1.0 / d)
    cp, rv, nz = C0.arrays()
    assert np.all(rv[idg - 1] == np.arange(1, n + 1)) and np.all((idg >= cp[:-1]) & (idg < cp[1:]))
    # a matrix without a stored diagonal entry in column 3: Jacobi gives Inf there, ILU0 has no idiag
    M2 = M.tolil()
    M2[2, 2] = 0.0
    M2 = sp.csc_matrix(M2)
    M2.eliminate_zeros()
    M2.sort_indices()
    C2 = orc.CSC(n, n, M2.indptr + 1, M2.indices + 1, M2.data)
    assert np.isinf(C2.jacobi()[2])
    with pytest.raises(ValueError):
        C2.ilu0()
