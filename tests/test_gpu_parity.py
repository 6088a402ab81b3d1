"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.

Integer structure (colptr/rowval) must match bit for bit; Float64 nzval must match BITWISE too:
the sort is stable and the fold runs left to right, so the device reproduces the reference's
accumulation order (the north star allows 2 ulp; the tests hold the stricter bar).
"""
import sys

import numpy as np
import pytest

import golden_util as gu
import sharded_model   # (test infrastructure: the Python model of the sharded flush over torch.distributed)
from refmodel import assert_csc_equal, bits, check_julia_invariants

pytestmark = pytest.mark.gpu

SET, UPDATE, RAW = 0, 1, 2


def hip_arrays(A):
    return A.sparse().arrays()


# ------------------------------------------------------------------ golden fixtures
@pytest.mark.parametrize("name", gu.STREAMS)
def test_golden_streams(esp, name):
    fx = gu.load(name)
    hashes = []
    for q, got, want, rebuilt in gu.replay(
            fx, esp.ExtendableSparseMatrix,
            lambda A, k, I, J, V: A.append(0, I, J, V, kinds=k),
            lambda A: (hashes.append(A.phash), A.flush(), hashes.append(A.phash)),
            hip_arrays):
        assert_csc_equal(got, want, "%s flush %d" % (name, q))
        check_julia_invariants(int(fx["m"]), int(fx["n"]), *got)
        # phash changes iff the CSC was rebuilt (extendable.jl:249-252)
        assert (hashes[-1] != hashes[-2]) == bool(rebuilt)


@pytest.mark.parametrize("dims", [(100, 1, 1), (10, 10, 1), (5, 5, 5)])
@pytest.mark.parametrize("mode", [0, 1])
def test_golden_fdrand_small(esp, dims, mode):
    fx = gu.load("fdrand_small")
    tag = "fd_%dx%dx%d_m%d" % (*dims, mode)
    want = (fx[tag + "_colptr"], fx[tag + "_rowval"], fx[tag + "_nzval"])
    A = esp.fdrand(*dims, rand_mode=mode, seed=0x5EED0002)          # device generator
    assert_csc_equal(hip_arrays(A), want, tag + " device")
    for upd in (esp.fdrand_module.update_updateindex, esp.fdrand_module.update_rawupdateindex,
                esp.fdrand_module.update_pluseq):                   # test_fdrand.jl:29-53
        if upd is esp.fdrand_module.update_pluseq and dims != (5, 5, 5):
            continue                                                # flush-per-getindex: keep it small
        B = esp.fdrand(*dims, rand_mode=mode, seed=0x5EED0002, update=upd)
        assert_csc_equal(hip_arrays(B), want, tag + " host loop")


@pytest.mark.parametrize("mode", [0, 1])
def test_golden_fdrand_30(esp, mode):
    """BASELINE config 1 (30^3) on the device path vs the committed digest."""
    d = gu.digests()["fd_30x30x30_m%d" % mode]
    A = esp.fdrand(30, 30, 30, rand_mode=mode, seed=0x5EED0002)
    arrs = hip_arrays(A)
    assert len(arrs[1]) == 183600
    assert gu.digest(*arrs) == d["csc"]


@pytest.mark.parametrize("dim,npd", [(2, 32), (3, 10)])
def test_golden_fem(esp, dim, npd):
    fx = gu.load("fem_small")
    tag = "fem%dd_%d" % (dim, npd)
    nn = npd ** dim
    A = esp.ExtendableSparseMatrix(nn, nn)
    A.generate_fem(dim, npd, seed=0x5EED0004, order_mode=1)
    A.flush()
    assert_csc_equal(hip_arrays(A), (fx[tag + "_colptr"], fx[tag + "_rowval"], fx[tag + "_nzval"]), tag)


@pytest.mark.parametrize("late", [False, True])
def test_group3_total_early_or_late(esp, monkeypatch, late):
    """group3_k publishes a RAWUPDATE segment's total right after the sort (one record per distinct row; the wave that counts
    its columns last publishes) -- or, esp_debug_force_path(36), after the fold as before: the same CSC either way (FEM fixtures)."""
    LATE = 36 if late else 0
    fx = gu.load("fem_small")
    for dim, npd in ((2, 32), (3, 10)):
        tag = "fem%dd_%d" % (dim, npd)
        nn = npd ** dim
        A = esp.ExtendableSparseMatrix(nn, nn)
        A.debug_force_path(LATE)
        A.generate_fem(dim, npd, seed=0x5EED0004, order_mode=1)
        A.flush()
        assert_csc_equal(hip_arrays(A), (fx[tag + "_colptr"], fx[tag + "_rowval"], fx[tag + "_nzval"]), tag)
    # a mesh large enough for many segments and look-back groups: device against device (early against late is the point)
    npd = 700
    B = esp.ExtendableSparseMatrix(npd * npd, npd * npd)
    B.debug_force_path(LATE)
    B.generate_fem(2, npd, seed=11, order_mode=1)
    B.flush()
    assert B.debug_last_local_small() == 2
    d = gu.digest(*hip_arrays(B))
    C2 = esp.ExtendableSparseMatrix(npd * npd, npd * npd)
    C2.generate_fem(2, npd, seed=11, order_mode=1)
    C2.flush()
    assert gu.digest(*hip_arrays(C2)) == d


# ------------------------------------------------------------------ reference test-suite, on the device
def test_updates_nnz_trace(esp):
    """test/test_updates.jl:10-25."""
    A = esp.ExtendableSparseMatrix(10, 10)
    assert A.nnz() == 0
    A[1, 3] = 5
    A.updateindex("+", 6.0, 4, 5)
    A.updateindex("+", 0.0, 2, 3)
    assert A.nnz() == 2
    A.rawupdateindex("+", 0.0, 2, 3)
    assert A.nnz() == 3
    A.dropzeros()
    assert A.nnz() == 2
    A.rawupdateindex("+", 0.1, 2, 3)
    assert A.nnz() == 3
    A.dropzeros()
    assert A.nnz() == 3


def _assembly(esp, orc, m, n, xnnz, nsplice, seed):
    """test/test_assembly.jl:6-35 with updateindex! (== `+=`) on the device."""
    rng = np.random.default_rng(seed)
    A = esp.ExtendableSparseMatrix(m, n)
    O = orc.ExtendableSparseMatrix(m, n)
    for _ in range(nsplice):
        I = rng.integers(1, m + 1, xnnz)
        J = rng.integers(1, n + 1, xnnz)
        a = 1.0 + rng.random(xnnz)
        A.append(UPDATE, I, J, a)
        O.apply(np.full(xnnz, orc.KIND_PLUSEQ, np.uint8), I, J, a)
        A.flush()
        O.flush()
        got = hip_arrays(A)
        check_julia_invariants(m, n, *got)
        assert_csc_equal(got, O.arrays())


@pytest.mark.parametrize("m,n,xnnz,nsplice", [
    (10, 10, 5, 1), (100, 100, 500, 2), (1000, 1000, 5000, 3),
    (20, 10, 5, 1), (200, 100, 500, 2), (2000, 1000, 5000, 3),
    (10, 20, 5, 1), (100, 200, 500, 2), (1000, 2000, 5000, 3),
])
def test_assembly_fixed_shapes(esp, orc, m, n, xnnz, nsplice):
    _assembly(esp, orc, m, n, xnnz, nsplice, seed=m * 7 + n)


def test_assembly_random_shapes(esp, orc):
    rng = np.random.default_rng(4321)
    for _ in range(10):
        m, n, z = (int(rng.integers(1, 10001)) for _ in range(3))
        _assembly(esp, orc, m, n, z, int(rng.integers(1, 6)), seed=z)


def test_pluseq_through_getindex(esp, orc):
    """`A[i,j] += v` (getindex + setindex!, docs/src/example.md:166-177): flush-then-lookup path."""
    rng = np.random.default_rng(3)
    m, n = 12, 9
    A = esp.ExtendableSparseMatrix(m, n)
    O = orc.ExtendableSparseMatrix(m, n)
    for _ in range(60):
        i, j, v = int(rng.integers(1, m + 1)), int(rng.integers(1, n + 1)), float(rng.standard_normal())
        A[i, j] = A[i, j] + v
        O.apply(np.array([orc.KIND_PLUSEQ], np.uint8), [i], [j], [v])
    assert_csc_equal(hip_arrays(A), O.arrays())


def test_mixed_kinds_and_zeros(esp, orc):
    rng = np.random.default_rng(7)
    m, n = 37, 23
    A = esp.ExtendableSparseMatrix(m, n)
    O = orc.ExtendableSparseMatrix(m, n)
    pool = np.array([0.0, -0.0, 1.5, -1.5, 1e-300, 3.25, -7.0])
    for splice in range(5):
        cnt = 700
        kinds = rng.integers(0, 3, cnt).astype(np.uint8)
        I = rng.integers(1, m + 1, cnt)
        J = rng.integers(1, n + 1, cnt)
        V = np.where(rng.random(cnt) < 0.4, rng.choice(pool, cnt), rng.standard_normal(cnt))
        A.append(0, I, J, V, kinds=kinds)
        O.apply(kinds, I, J, V)
        if splice % 2 == 0:
            A.flush()
            O.flush()
            assert_csc_equal(hip_arrays(A), O.arrays(), "splice %d" % splice)
    assert_csc_equal(hip_arrays(A), O.arrays())


@pytest.mark.parametrize("keys8", [False, True])
def test_host_append_key_formats(esp, orc, monkeypatch, keys8):
    """esp_append_host of ONE kind crosses PCIe with six-byte keys (low 32 bits + next 16, no kind; unpack6_k on the device) when
    row + column bits fit 48, else -- or with esp_debug_force_path(38) -- with packed eight-byte keys: the same CSC, bit for bit the
    oracle's, for every kind, op '-' (SET is not negated), Int32 index arrays, several chunks' worth of zeros and duplicates."""
    rng = np.random.default_rng(2024)
    m, n = 70001, 65537            # (17 + 17 key bits: the high sixteen of a six-byte key are in use)
    pool = np.array([0.0, -0.0, 2.5, -2.5, 1e-300])
    for kind in (SET, UPDATE, RAW):
        for op in ("+", "-"):
            A = esp.ExtendableSparseMatrix(m, n)
            if keys8:
                A.debug_force_path(38)
            O = orc.ExtendableSparseMatrix(m, n)
            for splice, i32 in enumerate((False, True, False)):
                cnt = 60000
                I = rng.integers(1, m + 1, cnt)
                J = rng.integers(1, n + 1, cnt)
                J[: cnt // 3] = rng.integers(1, 40, cnt // 3)          # (duplicates: a few dense columns)
                I[: cnt // 3] = rng.integers(1, 60, cnt // 3)
                V = np.where(rng.random(cnt) < 0.3, rng.choice(pool, cnt), rng.standard_normal(cnt))
                if i32:
                    A.append(kind, I.astype(np.int32), J.astype(np.int32), V, op=op)
                else:
                    A.append(kind, I, J, V, op=op)
                Vo = -V if (op == "-" and kind != SET) else V
                O.apply(np.full(cnt, kind, np.uint8), I, J, Vo)
                if splice == 1:
                    A.flush()
                    O.flush()
            assert_csc_equal(hip_arrays(A), O.arrays(), "kind %d op %s keys8 %s" % (kind, op, keys8))
    # an index outside the matrix: BoundsError, nothing appended (both formats check on the host)
    B = esp.ExtendableSparseMatrix(100, 100)
    with pytest.raises(Exception):
        B.append(UPDATE, np.array([1, 101]), np.array([1, 1]), np.array([1.0, 2.0]))
    assert B.nnznew() == 0


def test_per_entry_calls_equal_batch(esp, orc):
    rng = np.random.default_rng(11)
    m, n = 40, 50
    A = esp.ExtendableSparseMatrix(m, n)
    O = orc.ExtendableSparseMatrix(m, n)
    for _ in range(2000):
        k = int(rng.integers(0, 4))
        i, j = int(rng.integers(1, m + 1)), int(rng.integers(1, n + 1))
        v = float(rng.choice([0.0, 1.0, -2.5, rng.standard_normal()]))
        if k == 0:
            A[i, j] = v
            O[i, j] = v
        elif k == 1:
            A.updateindex("+", v, i, j)
            O.updateindex(orc.OP_ADD, v, i, j)
        elif k == 2:
            A.rawupdateindex("+", v, i, j)
            O.rawupdateindex(orc.OP_ADD, v, i, j)
        else:
            A.updateindex("-", v, i, j)
            O.updateindex(orc.OP_SUB, v, i, j)
    assert_csc_equal(hip_arrays(A), O.arrays())


def test_bounds_error(esp):
    A = esp.ExtendableSparseMatrix(5, 4)
    for (i, j) in [(0, 1), (6, 1), (1, 0), (1, 5)]:
        with pytest.raises(IndexError):
            A.updateindex("+", 1.0, i, j)
        with pytest.raises(IndexError):
            A[i, j] = 1.0
        with pytest.raises(IndexError):
            A[i, j]
        with pytest.raises(IndexError):   # device-side check of a bulk batch: nothing is committed
            A.append(UPDATE, [1, i], [1, j], [1.0, 1.0])
    assert A.nnznew() == 0 and A.nnz() == 0


def test_flush_gate_and_phash(esp):
    A = esp.ExtendableSparseMatrix(6, 6)
    assert A.phash == 0
    A.updateindex("+", 0.0, 2, 2)
    A[3, 3] = 0.0
    A.flush()
    assert A.phash == 0 and A.nnz() == 0
    A.rawupdateindex("+", 0.0, 2, 2)
    A.flush()
    h = A.phash
    assert h != 0 and A.nnz() == 1
    A.updateindex("+", 2.0, 2, 2)
    A.flush()
    assert A.phash == h and A[2, 2] == 2.0
    A.reset()
    assert A.nnz() == 0 and A.phash == h


def test_pattern_hash_matches_oracle_formula(esp, orc):
    A = esp.fdrand(6, 5, 4, rand_mode=1)
    cp, rv, nz = hip_arrays(A)
    assert A.phash == orc.CSC(120, 120, cp, rv, nz).pattern_hash()


def _sprand(rng, m, n, d):
    import scipy.sparse as sp
    S = sp.random(m, n, density=d, format="csc", random_state=rng, dtype=np.float64)
    S.sort_indices()
    return S


def test_csc_plus_buffer_is_2csc_and_round_trip(esp):
    """test_operations.jl:8-13 and test_constructors.jl:26-31 with the device buffer."""
    rng = np.random.default_rng(5)
    for _ in range(6):
        m, n = int(rng.integers(1, 600)), int(rng.integers(1, 600))
        S = _sprand(rng, m, n, 0.3 * rng.random())
        csc = esp.SparseMatrixCSC(m, n, S.indptr + 1, S.indices + 1, S.data)
        I, J, V = csc.findnz()
        x = esp.SparseMatrixHIPCOO(m, n)          # SparseMatrixLNK(csc): setindex! of every entry
        x.append(SET, I, J, V)
        two = csc + x
        assert two.pattern_equal(csc) and np.array_equal(bits(two.nzval), bits(2 * csc.nzval))
        y = esp.SparseMatrixHIPCOO(m, n)
        y.append(SET, I, J, V)
        back = y + esp.SparseMatrixCSC(m, n)
        assert back == csc


def test_generic_wrapper_vs_oracle(esp, orc):
    """GenericExtendableSparseMatrixCSC{SparseMatrixHIPCOO} (host routing) == ExtendableSparseMatrix."""
    rng = np.random.default_rng(17)
    m, n = 30, 45
    G = esp.GenericExtendableSparseMatrixCSC(m, n)
    O = orc.ExtendableSparseMatrix(m, n)
    for rnd in range(3):
        for _ in range(400):
            k = int(rng.integers(0, 3))
            i, j = int(rng.integers(1, m + 1)), int(rng.integers(1, n + 1))
            v = float(rng.choice([0.0, 2.0, rng.standard_normal()]))
            if k == 0:
                G[i, j] = v
                O[i, j] = v
            elif k == 1:
                G.updateindex("+", v, i, j)
                O.updateindex(orc.OP_ADD, v, i, j)
            else:
                G.rawupdateindex("+", v, i, j)
                O.rawupdateindex(orc.OP_ADD, v, i, j)
        # getindex of pending positions = the buffer's own lookup (the ordered fold of the pending calls on the
        # device, esp_pending_getindex), of stored ones = the CSC's (genericextendablesparsematrixcsc.jl:57-66)
        for _ in range(40):
            i, j = int(rng.integers(1, m + 1)), int(rng.integers(1, n + 1))
            assert bits(np.array([G[i, j]])) == bits(np.array([O[i, j]])), (rnd, i, j)
        G.flush()
        O.flush()
        assert_csc_equal(G.arrays(), O.arrays(), "round %d" % rnd)


def test_pending_getindex_and_release(esp, orc):
    """getindex(buffer,i,j) on a large pending batch (also a bucket-ordered one: 4-byte keys are expanded first), the
    `A[i,j] += v` idiom through it, bounds; esp_release_buffers leaves a valid empty buffer."""
    n = 48
    N = n ** 3
    X = esp.SparseMatrixHIPCOO(N, N)
    A = esp.ExtendableSparseMatrix(N, N)           # device generator into the buffer: X's API on the same handle
    X._d = A._d
    A.generate_fdrand(n, n, n, seed=3, rand_mode=1)
    O = orc.fdrand(n, n, n, rand_mode=1, seed=3, style=orc.KIND_UPDATE)    # (flushed: the fold of the same calls)
    rng = np.random.default_rng(8)
    for _ in range(25):
        l = int(rng.integers(1, N + 1))
        for (i, j) in ((l, l), (l, min(N, l + 1)), (min(N, l + n), l), (l, max(1, l - 7 * n))):
            assert bits(np.array([X[i, j]])) == bits(np.array([O[i, j]])), (i, j)
    with pytest.raises((esp.BoundsError, IndexError)):
        X[0, 1]
    A.flush()
    assert_csc_equal(hip_arrays(A), O.arrays())
    # += through the buffer lookup on a small matrix (SET after a read of the pending value)
    Y = esp.GenericExtendableSparseMatrixCSC(6, 7)
    Oy = orc.ExtendableSparseMatrix(6, 7)
    for t in range(200):
        i, j, v = int(rng.integers(1, 7)), int(rng.integers(1, 8)), float(rng.standard_normal())
        Y[i, j] = Y[i, j] + v
        Oy.apply(np.array([orc.KIND_PLUSEQ], np.uint8), [i], [j], [v])
    assert_csc_equal(Y.arrays(), Oy.arrays())
    # release: memory gone, handle still a valid empty buffer
    Z = esp.SparseMatrixHIPCOO(50, 60)
    Z.append(UPDATE, [1, 2, 50], [3, 60, 1], [1.0, 2.0, 3.0])
    assert Z.nnz() == 3
    Z.release()
    assert Z.nnz() == 0
    Z.updateindex("+", 4.0, 7, 8)
    C1 = Z + esp.SparseMatrixCSC(50, 60)
    assert C1.nnz() == 1 and C1[7, 8] == 4.0


@pytest.mark.parametrize("n,p,per_round", [(60, 3, 900), (400, 10, 20000), (1500, 20, 60000)])
def test_mt_wrapper_vs_oracle(esp, orc, n, p, per_round):
    """GenericMTExtendableSparseMatrixCSC{SparseMatrixHIPCOO}: test_parallel.jl:18-26 style, np = 3, 10, 20 partitions
    (test_parallel.jl:41,74).  flush! = Base.sum(xmatrices, csc) as ONE esp_flush_sum: bitwise the oracle's successive
    csc + buffer merges, over three rounds (new positions, then mostly hits on the host copy + a few new ones)."""
    rng = np.random.default_rng(19)
    M = esp.GenericMTExtendableSparseMatrixCSC(n, n, p)
    O = orc.MTExtendableSparseMatrix(n, n, p)
    with pytest.raises(RuntimeError):
        M[1, 1] = 1.0
    for rnd in range(3):
        I = rng.integers(1, n + 1, per_round)
        J = np.minimum(n, np.maximum(1, I + rng.integers(-20, 21, per_round)))
        T = rng.integers(1, p + 1, per_round)
        V = rng.standard_normal(per_round)
        for i, j, tid, v in zip(I.tolist(), J.tolist(), T.tolist(), V.tolist()):
            M.rawupdateindex("+", v, i, j, tid)
            O.rawupdateindex(orc.OP_ADD, v, i, j, tid)
        M.flush()
        O.flush()
        assert_csc_equal(M.arrays(), O.arrays(), "round %d" % rnd)
        assert M.nnznew() == 0


def test_flush_sum_and_values_only_transfers(esp, orc):
    """esp_flush_sum / esp_set_nzval / esp_get_nzval directly: Base.sum(xs, csc) with a handle that keeps the CSC between
    flushes.  Buffers with SET / UPDATE / RAWUPDATE calls (each folds by itself: sparsematrixdilnkc.jl:397-435), empty
    buffers, a round in which no position is new (values only come back, the pattern arrays are shared), nonzeros(A) .= 0 on
    the host in between (test_parallel.jl:71-92: values only go up), and a csc the handle has never seen (full upload)."""
    rng = np.random.default_rng(23)
    m, n, p = 700, 900, 6
    home = esp.SparseMatrixHIPCOO(m, n)
    xs = [esp.SparseMatrixHIPCOO(m, n) for _ in range(p)]
    csc = esp.SparseMatrixCSC(m, n)
    O = orc.CSC(m, n)
    pos = (rng.integers(1, m + 1, 5000), rng.integers(1, n + 1, 5000))
    for rnd in range(5):
        lnks = []
        for t, x in enumerate(xs):
            L = orc.SparseMatrixLNK(m, n)
            lnks.append(L)
            if t == 2 or (rnd == 3 and t % 2):
                continue                                      # an empty buffer
            cnt = 4000
            pick = rng.integers(0, 5000, cnt)
            I, J = pos[0][pick], pos[1][pick]
            if rnd in (0, 4):                                 # new positions as well
                I = np.where(rng.random(cnt) < 0.3, rng.integers(1, m + 1, cnt), I)
            elif rnd >= 1:
                keep = np.array([csc.findindex(int(i), int(j)) > 0 for i, j in zip(I, J)])
                I, J = I[keep], J[keep]                       # hits only: the pattern stays
            V = np.where(rng.random(len(I)) < 0.1, 0.0, rng.standard_normal(len(I)))
            K = rng.choice(np.array([0, 1, 2], np.uint8), len(I))
            x.append(0, I, J, V, kinds=K)
            for k, i, j, v in zip(K.tolist(), I.tolist(), J.tolist(), V.tolist()):
                if k == 0:
                    L[i, j] = v
                elif k == 1:
                    L.updateindex(orc.OP_ADD, v, i, j)
                else:
                    L.rawupdateindex(orc.OP_ADD, v, i, j)
        before = csc
        out = esp.SparseMatrixHIPCOO.sum(xs, csc, home=home)
        for L in lnks:
            if L.nnz() > 0:
                O = L + O                                     # (sparse! over csc, x1, x2, ...: successive csc + buffer)
        assert_csc_equal(out.arrays(), O.arrays(), "round %d" % rnd)
        assert all(x.nnz() == 0 for x in xs)
        if rnd in (1, 2, 3):
            assert out.colptr is before.colptr and out.rowval is before.rowval    # (values only came back)
        csc = out
        if rnd == 1:                                          # nonzeros(A) .= 0: only the values travel up
            csc.nzval[:] = 0.0
            O = orc.CSC(m, n, *[a.copy() for a in (O.arrays()[0], O.arrays()[1], np.zeros(O.nnz()))])
        if rnd == 2:                                          # a matrix the handle has never seen
            csc = esp.SparseMatrixCSC(m, n, csc.colptr.copy(), csc.rowval.copy(), csc.nzval.copy())
    # x + csc on one buffer keeps its result attached as well
    x = xs[0]
    x.updateindex("+", 2.5, int(pos[0][0]), int(pos[1][0]))
    L = orc.SparseMatrixLNK(m, n)
    L.updateindex(orc.OP_ADD, 2.5, int(pos[0][0]), int(pos[1][0]))
    r1 = x + csc
    O = L + O
    assert_csc_equal(r1.arrays(), O.arrays(), "x + csc")
    x.updateindex("+", -1.0, int(pos[0][1]), int(pos[1][1]))
    L = orc.SparseMatrixLNK(m, n)
    L.updateindex(orc.OP_ADD, -1.0, int(pos[0][1]), int(pos[1][1]))
    r2 = x + r1
    O = L + O
    assert_csc_equal(r2.arrays(), O.arrays(), "x + (x + csc)")
    assert r2.colptr is r1.colptr


@pytest.mark.parametrize("shape", ["bands", "everywhere"])
def test_flush_sum_batched_and_one_by_one(esp, orc, shape):
    """esp_flush_sum over buffers of per-entry calls (not element batches): the folds as ONE flush of a scratch matrix whose columns
    are the buffers' occupied column ranges side by side (round 6) against every buffer's own flush (a test hook on the destination
    switches the batched form off) against the oracle's successive csc + buffer.  Buffers in bands of columns that overlap their
    neighbours' / spread over the whole matrix, SET / UPDATE / RAWUPDATE calls, repeated positions, explicit zeros, an empty buffer,
    a stored matrix that some calls hit; two rounds (the second over the scratch handle kept from the first)."""
    import ctypes as C
    rng = np.random.default_rng(61)
    m, n, p = 5000, 60000, 7
    O = orc.CSC(m, n)
    homes = [esp.SparseMatrixHIPCOO(m, n), esp.SparseMatrixHIPCOO(m, n)]
    homes[1]._d.ck(homes[1]._d.lib.esp_debug_force_path(homes[1]._d.h, 31))     # (any hook on the destination: one by one)
    cscs = [esp.SparseMatrixCSC(m, n), esp.SparseMatrixCSC(m, n)]
    for rnd in range(2):
        lnks, streams = [], []
        for t in range(p):
            L = orc.SparseMatrixLNK(m, n)
            lnks.append(L)
            if t == 3:
                streams.append(None)                      # an empty buffer
                continue
            cnt = 30000 + 1000 * t
            if shape == "bands":
                lo, hi = 1 + t * n // p - 500 * (t > 0), (t + 1) * n // p + 500 * (t + 1 < p)
                J = np.sort(rng.integers(lo, hi + 1, cnt))
            else:
                J = rng.integers(1, n + 1, cnt)
            I = rng.integers(1, 40, cnt) + (J % (m - 50))     # (few rows per column: repeated positions)
            V = np.where(rng.random(cnt) < 0.05, 0.0, rng.standard_normal(cnt))
            K = rng.choice(np.array([0, 1, 2], np.uint8), cnt, p=[0.1, 0.6, 0.3])
            streams.append((I, J, V, K))
            for k, i, j, v in zip(K.tolist(), I.tolist(), J.tolist(), V.tolist()):
                if k == 0:
                    L[i, j] = v
                elif k == 1:
                    L.updateindex(orc.OP_ADD, v, i, j)
                else:
                    L.rawupdateindex(orc.OP_ADD, v, i, j)
        for L in lnks:
            if L.nnz() > 0:
                O = L + O
        for which, home in enumerate(homes):
            xs = [esp.SparseMatrixHIPCOO(m, n) for _ in range(p)]
            for x, st in zip(xs, streams):
                if st is not None:
                    x.append(0, st[0], st[1], st[2], kinds=st[3])
            cscs[which] = esp.SparseMatrixHIPCOO.sum(xs, cscs[which], home=home)
            flag = C.c_int32(-1)
            home._d.ck(home._d.lib.esp_debug_last_sum_batched(home._d.h, C.byref(flag)))
            assert flag.value == (1 if which == 0 else 0), (which, flag.value)
            assert_csc_equal(cscs[which].arrays(), O.arrays(), "round %d %s" % (rnd, "batched" if which == 0 else "one by one"))
            for x in xs:
                x.close() if hasattr(x, "close") else None


def test_device_consumer_hand_off_and_external_stream(esp, orc):
    """esp_csc_device (the hand-off to consumers that stay on the GPU), esp_get_nzval and esp_set_stream: an assembly on a
    caller-made stream, the device CSC read straight from the pointers the library hands out."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")          # (the runtime the library itself is linked against)
    n = 24
    N = n ** 3
    A = esp.ExtendableSparseMatrix(N, N)
    d = A._d
    stream = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(stream)) == 0
    d.ck(d.lib.esp_set_stream(d.h, stream))
    A.generate_fdrand(n, n, n, seed=0x5EED0002, rand_mode=1)
    A.flush()
    want = orc.fdrand(n, n, n, rand_mode=1, seed=0x5EED0002, style=orc.KIND_UPDATE).arrays()
    Z = A.nnz()
    assert Z == len(want[1])
    pc, pr, pv = C.c_void_p(), C.c_void_p(), C.c_void_p()
    d.ck(d.lib.esp_csc_device(d.h, C.byref(pc), C.byref(pr), C.byref(pv)))
    A.synchronize()
    cp, rv, nz = np.empty(N + 1, np.int64), np.empty(Z, np.int64), np.empty(Z, np.float64)
    for dst, src in ((cp, pc), (rv, pr), (nz, pv)):
        assert hip.hipMemcpy(C.c_void_p(dst.ctypes.data), src, C.c_size_t(dst.nbytes), 2) == 0     # hipMemcpyDeviceToHost
    assert_csc_equal((cp, rv, nz), want, "device pointers")
    nz2 = np.empty(Z, np.float64)
    d.ck(d.lib.esp_get_nzval(d.h, C.c_void_p(nz2.ctypes.data)))
    assert np.array_equal(bits(nz2), bits(want[2]))
    # a re-assembly on the same external stream: the pointers stay valid (no new position), the values double
    A.generate_fdrand(n, n, n, seed=0x5EED0002, rand_mode=1)
    A.flush()
    d.ck(d.lib.esp_get_nzval(d.h, C.c_void_p(nz2.ctypes.data)))
    O = orc.fdrand(n, n, n, rand_mode=1, seed=0x5EED0002, style=orc.KIND_UPDATE)
    I, J, V = orc.fdrand_stream(n, n, n, rand_mode=1, seed=0x5EED0002)
    O.apply(np.full(len(I), UPDATE, np.uint8), I, J, V)
    O.flush()
    assert np.array_equal(bits(nz2), bits(O.arrays()[2]))
    del A
    d.close()
    assert hip.hipStreamDestroy(stream) == 0


def test_fdrand_stream_and_reassembly(esp, orc):
    """SURVEY 3.2 / config 3 in the small: re-assembly hits the CSC in place, then +new entries."""
    nx, ny, nz = 9, 8, 7
    N = nx * ny * nz
    A = esp.fdrand(nx, ny, nz, rand_mode=1, seed=5)
    O = orc.fdrand(nx, ny, nz, rand_mode=1, seed=5, style=orc.KIND_UPDATE)
    assert_csc_equal(hip_arrays(A), O.arrays())
    h0 = A.phash
    esp.fdrand_device_(A, nx, ny, nz, rand_mode=1, seed=6)          # all hits
    O.fdrand(nx, ny, nz, rand_mode=1, seed=6, style=orc.KIND_UPDATE)
    assert_csc_equal(hip_arrays(A), O.arrays())
    assert A.phash == h0
    # second-neighbour pairs in x: new positions, merged with the existing CSC
    l = np.array([g + 1 for g in range(N) if g % nx < nx - 2], np.int64)
    rng = np.random.default_rng(9)
    v = rng.random(len(l))
    I = np.concatenate([l, l + 2])
    J = np.concatenate([l + 2, l])
    V = np.concatenate([v, v])
    A.generate_fdrand(nx, ny, nz, seed=7, rand_mode=1)
    A.append(UPDATE, I, J, V)
    Io, Jo, Vo = orc.fdrand_stream(nx, ny, nz, rand_mode=1, seed=7)
    O.apply(np.full(len(Io), UPDATE, np.uint8), Io, Jo, Vo)
    O.apply(np.full(len(I), UPDATE, np.uint8), I, J, V)
    A.flush()
    O.flush()
    assert_csc_equal(hip_arrays(A), O.arrays())
    assert A.phash != h0 and A.nnz() == orc.fdrand_nnz(nx, ny, nz) + len(I)


def test_device_generator_stream_is_the_reference_stream(esp, orc):
    """The on-device fdrand stream equals the sequential stream entry for entry (order included):
    feed it through a 1-entry-per-key matrix so every value is observable."""
    nx, ny, nz = 7, 6, 5
    N = nx * ny * nz
    I, J, V = orc.fdrand_stream(nx, ny, nz, rand_mode=1, seed=0xABC)
    A = esp.ExtendableSparseMatrix(N, N)
    A.generate_fdrand(nx, ny, nz, seed=0xABC, rand_mode=1)
    B = esp.ExtendableSparseMatrix(N, N)
    B.append(UPDATE, I, J, V)
    assert A.nnznew() == len(I) == orc.fdrand_count(nx, ny, nz)
    A.flush()
    B.flush()
    assert_csc_equal(hip_arrays(A), hip_arrays(B))


def test_long_duplicate_runs_and_dense_column(esp, orc):
    """Segments far longer than a tile: 30000 updates of one entry + a dense column."""
    m, n = 5000, 64
    rng = np.random.default_rng(23)
    I = np.concatenate([np.full(30000, 17), rng.integers(1, m + 1, 40000), np.arange(1, m + 1)])
    J = np.concatenate([np.full(30000, 5), np.full(40000, 9), np.full(m, 33)])
    V = rng.standard_normal(len(I))
    kinds = rng.integers(0, 3, len(I)).astype(np.uint8)
    A = esp.ExtendableSparseMatrix(m, n)
    O = orc.ExtendableSparseMatrix(m, n)
    A.append(0, I, J, V, kinds=kinds)
    O.apply(kinds, I, J, V)
    assert_csc_equal(hip_arrays(A), O.arrays())
    assert A.debug_last_path() == 2        # runs longer than an LDS bucket take the general path


@pytest.mark.parametrize("force", [0, 2, 3, 4])
def test_both_pipelines_agree_with_oracle(esp, orc, force):
    """The LDS bucket path (1) and the general path (2) are both checked against the oracle."""
    rng = np.random.default_rng(29)
    m, n = 3000, 2500
    A = esp.ExtendableSparseMatrix(m, n)
    A.debug_force_path(force)
    O = orc.ExtendableSparseMatrix(m, n)
    for rnd in range(3):
        cnt = 60000
        kinds = rng.integers(0, 3, cnt).astype(np.uint8)
        I = rng.integers(1, m + 1, cnt)
        J = rng.integers(1, n + 1, cnt)
        hot = rng.random(cnt) < 0.3
        I[hot] = rng.integers(1, 40, hot.sum())
        J[hot] = rng.integers(1, 30, hot.sum())
        V = np.where(rng.random(cnt) < 0.2, 0.0, rng.standard_normal(cnt))
        A.append(0, I, J, V, kinds=kinds)
        O.apply(kinds, I, J, V)
        A.flush()
        O.flush()
        assert A.debug_last_path() == (2 if force == 2 else 1)
        assert_csc_equal(hip_arrays(A), O.arrays(), "round %d" % rnd)


def test_bucket_kernel_many_launches(esp, orc):
    """More segments than one launch takes: ticket counter and look-back state carry over."""
    rng = np.random.default_rng(41)
    m, n = 2000, 300000
    cnt = 1500000
    A = esp.ExtendableSparseMatrix(m, n)
    A.debug_force_path(4)
    O = orc.ExtendableSparseMatrix(m, n)
    I = rng.integers(1, m + 1, cnt)
    J = rng.integers(1, n + 1, cnt)
    V = rng.standard_normal(cnt)
    A.append(UPDATE, I, J, V)
    O.apply(np.full(cnt, UPDATE, np.uint8), I, J, V)
    assert_csc_equal(hip_arrays(A), O.arrays())
    assert A.debug_last_path() == 1


@pytest.mark.parametrize("per_col", [6, 30, 200])
def test_bucket_kernel_tiers(esp, orc, per_col):
    """Column runs of ~6 (register sorting network), ~30 and ~200 (radix tier: varying key bits only, one
    (col,row) group per thread in the fold) entries, with duplicates and SET/zero entries, all through the
    LDS bucket kernel."""
    rng = np.random.default_rng(per_col)
    m, n = 700, 2048
    cnt = per_col * n
    A = esp.ExtendableSparseMatrix(m, n)
    O = orc.ExtendableSparseMatrix(m, n)
    for rnd in range(2):
        kinds = rng.choice(np.array([0, 1, 1, 1, 2], np.uint8), cnt)
        I = rng.integers(1, m + 1, cnt)
        J = rng.integers(1, n + 1, cnt)
        dup = rng.random(cnt) < 0.3
        I[dup] = rng.integers(1, 9, dup.sum())
        V = np.where(rng.random(cnt) < 0.1, 0.0, rng.standard_normal(cnt))
        A.append(0, I, J, V, kinds=kinds)
        O.apply(kinds, I, J, V)
        A.flush()
        O.flush()
        assert A.debug_last_path() == 1
        assert_csc_equal(hip_arrays(A), O.arrays(), "round %d" % rnd)


def test_host_append_int32_indices_and_kinds(esp, orc):
    """esp_append_host_i32 (Ti = Int32 callers) and the host-side packing of esp_append_host: per-entry kinds, `-`, a
    BoundsError in a later chunk of a large batch (nothing is committed), against the oracle."""
    rng = np.random.default_rng(31)
    m, n, cnt = 1_000_003, 900_001, 600000       # (short columns: the oracle's list walks stay short)
    I = rng.integers(1, m + 1, cnt)
    J = rng.integers(1, n + 1, cnt)
    V = np.where(rng.random(cnt) < 0.05, 0.0, rng.standard_normal(cnt))
    K = rng.choice(np.array([0, 1, 2], np.uint8), cnt)
    O = orc.ExtendableSparseMatrix(m, n)
    O.apply(K, I, J, V)
    O.apply(np.full(cnt, UPDATE, np.uint8), I, J, -V)
    A = esp.ExtendableSparseMatrix(m, n)
    A.append(0, I.astype(np.int32), J.astype(np.int32), V, kinds=K)          # Int32 arrays go as they are
    A.append(UPDATE, I, J, V, op="-")
    assert A.nnznew() == 2 * cnt
    A.flush()
    assert_csc_equal(hip_arrays(A), O.arrays(), "int32 + int64 host appends")
    big = 9_000_000                                                          # three chunks of the staging area
    Ib = rng.integers(1, m + 1, big).astype(np.int32)
    Jb = rng.integers(1, n + 1, big).astype(np.int32)
    Vb = rng.standard_normal(big)
    Jb[8_500_000] = n + 1
    with pytest.raises(esp.BoundsError) as ei:
        A.append(UPDATE, Ib, Jb, Vb)
    assert "8500001" in str(ei.value) and A.nnznew() == 0
    Jb[8_500_000] = 1
    A.append(UPDATE, Ib, Jb, Vb)
    O.apply(np.full(big, UPDATE, np.uint8), Ib.astype(np.int64), Jb.astype(np.int64), Vb)
    A.flush()
    O.flush()
    assert_csc_equal(hip_arrays(A), O.arrays(), "large int32 batch")


def test_append_device_reuses_the_previous_run_lists(esp, orc):
    """A caller that repeats its stream (a time-stepping code): esp_append_device of one kind on an empty buffer uses the run
    lists of the previous assembly -- no count pass over the columns -- and the scatter kernel checks every tile against its
    list.  The same stream twice, new values at the same positions, then a changed stream of the same length (falls back,
    makes a plan of its own, which the next repetition uses), entries swapped inside a tile (same digits and counts: the
    lists still fit), a BoundsError in a repeated batch, force_path 31 (never); always the oracle's bits."""
    import torch
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()   # noqa: E731
    n = 64
    N = n ** 3
    I, J, V = orc.fdrand_stream(n, n, n, rand_mode=1, seed=0x5EED0002)
    E = len(I)
    rng = np.random.default_rng(41)

    def check(A, Ii, Jj, Vv, expect_reused, what):
        A.reset()
        A.append_device(UPDATE, dev(Ii), dev(Jj), dev(Vv))
        assert A.debug_last_plan_reused() == expect_reused, (what, A.debug_last_plan_reused())
        A.flush()
        assert A.debug_last_partition() == 4
        O = orc.ExtendableSparseMatrix(N, N)
        O.apply(np.full(len(Ii), UPDATE, np.uint8), Ii, Jj, Vv)
        assert_csc_equal(hip_arrays(A), O.arrays(), what)

    A = esp.ExtendableSparseMatrix(N, N)
    check(A, I, J, V, 0, "first assembly")
    check(A, I, J, V, 1, "the same stream again")
    check(A, I, J, rng.standard_normal(E), 1, "new values, same positions")
    # swapped inside a tile: the digits and counts of every tile stay what they were
    I2, J2, V2 = I.copy(), J.copy(), V.copy()
    for t0 in range(0, E - 4096, 4096 * 37):
        a, b = t0 + 5, t0 + 3000
        for X in (I2, J2, V2):
            X[a], X[b] = X[b], X[a]
    check(A, I2, J2, V2, 1, "entries swapped inside their tiles")
    # a different stream of the same length: two far-apart blocks trade places
    I3, J3, V3 = I.copy(), J.copy(), V.copy()
    blk = 50000
    for X in (I3, J3, V3):
        tmp = X[:blk].copy()
        X[:blk] = X[E // 2:E // 2 + blk]
        X[E // 2:E // 2 + blk] = tmp
    check(A, I3, J3, V3, 0, "another stream of the same length")
    check(A, I3, J3, V3, 1, "... repeated")
    J4 = J3.copy()
    J4[E - 7] = N + 1
    A.reset()
    with pytest.raises(esp.BoundsError):
        A.append_device(UPDATE, dev(I3), dev(J4), dev(V3))
    assert A.nnznew() == 0
    check(A, I3, J3, V3, 0, "after the refused batch")
    B = esp.ExtendableSparseMatrix(N, N)
    B.debug_force_path(31)
    check(B, I, J, V, 0, "never")
    check(B, I, J, V, 0, "never, again")
    # a flush that partitions by itself in between rewrites the tables: no reuse right after it
    check(A, I3, J3, V3, 1, "repeated once more")
    A.reset()
    A.append(UPDATE, I[:300000], J[:300000], V[:300000])          # (host append: packed keys, the flush partitions)
    A.flush()
    check(A, I3, J3, V3, 0, "after another stream's own partition")


def test_append_device_entry_point(esp, orc):
    """esp_append_device (triplets resident in GPU memory) by itself: on an empty buffer a pre-sorted batch of one kind is
    partitioned as it is appended (esp_debug_last_partition 4), an unsorted one or one with a kinds array is packed in
    stream order; op '-' negates; further batches behind the first; an out-of-range entry rejects its whole batch and
    leaves what was appended before; lengths are checked by the host mirror."""
    import torch
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()   # noqa: E731
    rng = np.random.default_rng(321)
    m, n, cnt = 300000, 200000, 3000000
    for case in ("sorted_update", "sorted_raw_minus", "unsorted", "kinds_array", "two_batches", "bounds"):
        A = esp.ExtendableSparseMatrix(m, n)
        O = orc.ExtendableSparseMatrix(m, n)
        I = rng.integers(1, m + 1, cnt)
        J = rng.integers(1, n + 1, cnt)
        V = rng.standard_normal(cnt)
        V[rng.random(cnt) < 0.05] = 0.0
        if case != "unsorted":
            J = np.sort(J)
        if case == "sorted_raw_minus":
            A.append_device(RAW, dev(I), dev(J), dev(V), op="-")
            O.apply(np.full(cnt, RAW, np.uint8), I, J, -V)
        elif case == "kinds_array":
            kinds = rng.integers(0, 3, cnt).astype(np.uint8)
            A.append_device(0, dev(I), dev(J), dev(V), kinds=dev(kinds))
            O.apply(kinds, I, J, V)
        else:
            A.append_device(UPDATE, dev(I), dev(J), dev(V))
            O.apply(np.full(cnt, UPDATE, np.uint8), I, J, V)
        if case == "two_batches":
            I2, J2, V2 = rng.integers(1, m + 1, 5000), np.sort(rng.integers(1, n + 1, 5000)), rng.standard_normal(5000)
            A.append_device(orc.KIND_SET, dev(I2), dev(J2), dev(V2))
            O.apply(np.full(5000, orc.KIND_SET, np.uint8), I2, J2, V2)
        if case == "bounds":
            bad = J.copy()
            bad[7] = n + 1
            with pytest.raises((esp.BoundsError, IndexError)):
                A.append_device(UPDATE, dev(I), dev(bad), dev(V))
            with pytest.raises(ValueError):
                A.append_device(UPDATE, dev(I[:10]), dev(J), dev(V))
        A.flush(), O.flush()
        if case in ("sorted_update", "sorted_raw_minus", "bounds"):
            assert A.debug_last_partition() == 4, (case, A.debug_last_partition())
        assert_csc_equal(hip_arrays(A), O.arrays(), case)
        # ... and once more over the stored pattern (routed), the same entry point
        A.append_device(UPDATE, dev(I), dev(J), dev(V))
        O.apply(np.full(cnt, UPDATE, np.uint8), I, J, V)
        A.flush(), O.flush()
        assert_csc_equal(hip_arrays(A), O.arrays(), case + " again")


def test_tail_partitioned_as_appended(esp, orc):
    """An append of one kind behind a producer's batch over a STORED pattern is partitioned as it comes (partition.hip,
    append_tail_partitioned): kinds UPDATE / SET / RAWUPDATE, op '-', a tail whose columns come in no sorted order (falls
    back to the packed append), a second append behind the tail (the partition's bookkeeping is dropped, the entries
    stay), and an out-of-range entry (the whole append is rejected, the batch stays pending) -- the oracle's bits."""
    import torch
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()   # noqa: E731  (esp_append_device: resident triplets)
    n = 96
    N = n ** 3
    rng = np.random.default_rng(123)
    Ib, Jb, Vb = orc.fdrand_stream(n, n, n, rand_mode=1, seed=81)
    l = np.array([g + 1 for g in range(N) if g % n < n - 2], np.int64)      # second neighbours in x: new positions
    It, Jt = np.concatenate([l, l + 2]), np.concatenate([l + 2, l])
    for case in ("update", "set", "raw", "minus", "unsorted", "two_appends", "bounds"):
        A = esp.ExtendableSparseMatrix(N, N)
        A.generate_fdrand(n, n, n, seed=80, rand_mode=1)
        A.flush()
        O = orc.fdrand(n, n, n, rand_mode=1, seed=80, style=orc.KIND_UPDATE)
        A.generate_fdrand(n, n, n, seed=81, rand_mode=1)                    # the batch (every entry hits)
        O.apply(np.full(len(Ib), UPDATE, np.uint8), Ib, Jb, Vb)
        Vt = rng.standard_normal(len(It))
        Vt[rng.random(len(It)) < 0.05] = 0.0
        I, J = It, Jt
        kind = {"set": orc.KIND_SET, "raw": orc.KIND_RAWUPDATE}.get(case, UPDATE)
        if case == "unsorted":
            q = rng.permutation(len(I))
            I, J, Vt = I[q], J[q], Vt[q]
        if case == "bounds":
            bad = I.copy()
            bad[len(bad) // 3] = N + 1
            with pytest.raises((esp.BoundsError, IndexError)):
                A.append_device(UPDATE, dev(bad), dev(J), dev(Vt))
            A.flush(), O.flush()                                              # (the batch alone)
            assert_csc_equal(hip_arrays(A), O.arrays(), case)
            continue
        if case == "minus":
            A.append_device(UPDATE, dev(I), dev(J), dev(Vt), op="-")
            O.apply(np.full(len(I), UPDATE, np.uint8), I, J, -Vt)
        else:
            A.append_device(kind, dev(I), dev(J), dev(Vt))
            O.apply(np.full(len(I), kind, np.uint8), I, J, Vt)
        if case == "two_appends":
            k = 5000
            I2, J2, V2 = rng.integers(1, N + 1, k), np.sort(rng.integers(1, N + 1, k)), rng.standard_normal(k)
            A.append(UPDATE, I2, J2, V2)
            O.apply(np.full(k, UPDATE, np.uint8), I2, J2, V2)
        A.flush(), O.flush()
        # (the batch by itself, then what came behind it: 8 = partitioned as it was appended, 6 = at the flush)
        assert A.debug_last_partition() == (8 if case in ("update", "set", "raw", "minus") else 6), (case, A.debug_last_partition())
        assert_csc_equal(hip_arrays(A), O.arrays(), case)


def test_routed_fold_short_columns_and_history(esp, orc):
    """The routed fold over SHORT stored columns (rows and values of a column fetched at once, merge in registers:
    local.hpp fold_run_short_csc) and the walk, chosen by what the handle's last flush over the pattern did (hits /
    mostly new positions): all-hit re-assembly, a flush of mostly new positions, then re-assembly twice (the first walks,
    the second takes the short-column fold again); kinds SET / UPDATE / RAWUPDATE mixed, a zero update on an absent
    position, and csc + buffer (ESP_FLUSH_PLUS: the stored value is the first operand).  A column that grew beyond 8
    entries walks."""
    n = 24
    N = n ** 3
    rng = np.random.default_rng(77)
    A = esp.ExtendableSparseMatrix(N, N)
    A.generate_fdrand(n, n, n, seed=61, rand_mode=1)
    A.flush()
    O = orc.fdrand(n, n, n, rand_mode=1, seed=61, style=orc.KIND_UPDATE)
    assert_csc_equal(hip_arrays(A), O.arrays())

    def both(kinds, I, J, V):
        A.append(0, I, J, V, kinds=kinds)
        O.apply(kinds, I, J, V)

    def reassemble(seed, mixed):
        I, J, V = orc.fdrand_stream(n, n, n, rand_mode=1, seed=seed)
        if mixed:
            kinds = rng.choice(np.array([UPDATE, orc.KIND_SET, orc.KIND_RAWUPDATE], np.uint8), len(I))
            V = np.where(rng.random(len(I)) < 0.1, 0.0, V)
            both(kinds, I, J, V)
        else:
            A.generate_fdrand(n, n, n, seed=seed, rand_mode=1)
            O.apply(np.full(len(I), UPDATE, np.uint8), I, J, V)

    reassemble(62, False)                                  # all hits: short-column fold, values fetched with the rows
    A.flush(), O.flush()
    assert_csc_equal(hip_arrays(A), O.arrays(), "all hits")
    k = N // 2                                             # mostly new positions, some hits, zero updates on absent positions
    In, Jn = rng.integers(1, N + 1, k), np.sort(rng.integers(1, N + 1, k))
    Vn = np.where(rng.random(k) < 0.2, 0.0, rng.standard_normal(k))
    both(np.full(k, UPDATE, np.uint8), In, Jn, Vn)
    A.flush(), O.flush()
    assert_csc_equal(hip_arrays(A), O.arrays(), "mostly new")
    for rnd, mixed in enumerate((False, True, True)):      # history says "new positions" once, then "hits" again
        reassemble(70 + rnd, mixed)
        if rnd == 2:                                       # ... with a few new positions among the hits
            both(np.full(50, UPDATE, np.uint8), rng.integers(1, N + 1, 50), np.sort(rng.integers(1, N + 1, 50)), rng.standard_normal(50))
        A.flush(), O.flush()
        assert_csc_equal(hip_arrays(A), O.arrays(), "re-assembly %d" % rnd)
    # csc + buffer over short columns (ESP_FLUSH_PLUS): the buffer is folded by itself, the stored value is the first operand
    cp, rv, nz = O.arrays()
    I, J, V = orc.fdrand_stream(n, n, n, rand_mode=1, seed=63)
    I, J, V = I[:60000], J[:60000], V[:60000]
    X = esp.SparseMatrixHIPCOO(N, N)
    L = orc.SparseMatrixLNK(N, N)
    X.append(UPDATE, I, J, V)
    for i, j, v in zip(I, J, V):
        L.updateindex(orc.OP_ADD, float(v), int(i), int(j))
    got = X + esp.SparseMatrixCSC(N, N, cp, rv, nz)
    want = L + orc.CSC(N, N, cp, rv, nz)
    assert_csc_equal(got.arrays(), want.arrays())


@pytest.mark.parametrize("force", [0, 17])
def test_reassembly_with_a_few_new_entries(esp, orc, force):
    """Re-assembly over the stored pattern: almost every segment of the bucket kernel emits nothing (and does
    not wait for its look-back chain); a handful of new positions far apart must still land at the right
    offsets, across long stretches of such segments.  Join: column-tiled (0) / merge-path (17)."""
    n = 48
    N = n ** 3
    A = esp.ExtendableSparseMatrix(N, N)
    A.debug_force_path(force)
    A.generate_fdrand(n, n, n, seed=41, rand_mode=1)
    A.flush()
    O = orc.fdrand(n, n, n, rand_mode=1, seed=41, style=orc.KIND_UPDATE)
    rng = np.random.default_rng(43)
    for rnd in range(3):
        I, J, V = orc.fdrand_stream(n, n, n, rand_mode=1, seed=50 + rnd)
        A.generate_fdrand(n, n, n, seed=50 + rnd, rand_mode=1)
        O.apply(np.full(len(I), UPDATE, np.uint8), I, J, V)
        if rnd > 0:                                   # new positions: 1, then 7, spread over the columns
            k = 1 if rnd == 1 else 7
            In = rng.integers(1, N + 1, k)
            Jn = np.sort(rng.choice(N, k, replace=False)) + 1
            Vn = rng.standard_normal(k)
            A.append(UPDATE, In, Jn, Vn)
            O.apply(np.full(k, UPDATE, np.uint8), In, Jn, Vn)
        A.flush()
        O.flush()
        assert A.debug_last_path() == 1
        assert_csc_equal(hip_arrays(A), O.arrays(), "round %d" % rnd)


def test_copy_is_independent(esp, orc):
    """Base.copy(ext) (extendable.jl:279-285): CSC and pending entries are copied; the two matrices then evolve
    independently (device-to-device copy of the handle)."""
    rng = np.random.default_rng(90)
    m, n, cnt = 400, 300, 20000
    A = esp.ExtendableSparseMatrix(m, n)
    O = orc.ExtendableSparseMatrix(m, n)
    I, J, V = rng.integers(1, m + 1, cnt), rng.integers(1, n + 1, cnt), rng.standard_normal(cnt)
    A.append(UPDATE, I, J, V)
    O.apply(np.full(cnt, UPDATE, np.uint8), I, J, V)
    A.flush()
    I2, J2, V2 = rng.integers(1, m + 1, 500), rng.integers(1, n + 1, 500), rng.standard_normal(500)
    A.append(UPDATE, I2, J2, V2)                       # pending at the time of the copy
    A.updateindex("+", 2.5, 7, 9)                      # (still in the staging chunk)
    B = A.copy()
    assert B.phash == A.phash and B.nnznew() == A.nnznew() == 501
    O.apply(np.full(500, UPDATE, np.uint8), I2, J2, V2)
    O.updateindex(orc.OP_ADD, 2.5, 7, 9)
    want = O.arrays()
    A[1, 1] = 99.0                                     # only A changes from here on
    assert_csc_equal(hip_arrays(B), want)
    O[1, 1] = 99.0
    assert_csc_equal(hip_arrays(A), O.arrays())
    assert_csc_equal(hip_arrays(B), want)


def test_staged_pushes_around_a_bulk_append(esp, orc):
    """Per-entry updates (staged in the caller's pinned chunk) before and after a bulk append that is larger
    than that chunk: the bulk path has its own staging area, the caller's chunk pointers stay valid."""
    rng = np.random.default_rng(88)
    m, n = 500, 400
    A = esp.ExtendableSparseMatrix(m, n)
    O = orc.ExtendableSparseMatrix(m, n)
    for i in range(100):
        A.updateindex("+", 1.0 + i, 1 + i % m, 1 + (7 * i) % n)
        O.updateindex(orc.OP_ADD, 1.0 + i, 1 + i % m, 1 + (7 * i) % n)
    cnt = 300000
    I = rng.integers(1, m + 1, cnt)
    J = rng.integers(1, n + 1, cnt)
    V = rng.standard_normal(cnt)
    A.append(UPDATE, I, J, V)
    O.apply(np.full(cnt, UPDATE, np.uint8), I, J, V)
    for i in range(100):
        A[1 + (3 * i) % m, 1 + i % n] = float(i)
        O[1 + (3 * i) % m, 1 + i % n] = float(i)
    assert_csc_equal(hip_arrays(A), O.arrays())
    # an out-of-range entry in the middle of a bulk batch rejects the whole batch, nothing is committed
    I2 = I.copy()
    I2[cnt // 2] = m + 1
    with pytest.raises((esp.BoundsError, IndexError)):
        A.append(UPDATE, I2, J, V)
    assert A.nnznew() == 0
    assert_csc_equal(hip_arrays(A), O.arrays())


def test_bucket_kernel_24_input_tier(esp, orc):
    """Column runs of exactly 20 entries: the first flush of a handle takes the radix tier, the
    following ones the kernel variant with the 24-input register tier (chosen from the longest run the
    previous flush met) -- over an existing CSC (ROUTED) and, after reset!, on a fresh matrix."""
    rng = np.random.default_rng(24)
    m, n, per = 900, 4096, 20
    cnt = per * n
    A = esp.ExtendableSparseMatrix(m, n)
    O = orc.ExtendableSparseMatrix(m, n)
    for rnd in range(4):
        if rnd == 2:
            A.reset()
            O.reset()
        kinds = rng.choice(np.array([0, 1, 1, 1, 2], np.uint8), cnt)
        J = rng.permutation(np.repeat(np.arange(1, n + 1), per))
        I = rng.integers(1, m + 1, cnt)
        dup = rng.random(cnt) < 0.4
        I[dup] = rng.integers(1, 7, dup.sum())
        V = np.where(rng.random(cnt) < 0.1, 0.0, rng.standard_normal(cnt))
        A.append(0, I, J, V, kinds=kinds)
        O.apply(kinds, I, J, V)
        A.flush()
        O.flush()
        assert A.debug_last_path() == 1
        assert_csc_equal(hip_arrays(A), O.arrays(), "round %d" % rnd)


@pytest.mark.parametrize("per_col", [20, 28, 50, 100, 120, 200, 250])
@pytest.mark.parametrize("force", [0, 24])
def test_bucket_kernel_group_tier(esp, orc, per_col, force):
    """Column runs of 17 .. 256 entries (P1 FEM: 24 per column in 2-D, 120 in 3-D): the group tier -- 2, 4, 8 or 16
    lanes per column, 16 keys each in registers, merged across lanes by DPP (local.hpp, group_sort) -- against the
    radix tier (force_path 24) and the oracle; mixed kinds, zeros, duplicates, columns of different lengths (some
    empty), first on a fresh matrix, then over the stored CSC, then with rows more than 2^18 apart (the tier's 32-bit
    sort keys do not apply: radix tier)."""
    rng = np.random.default_rng(per_col * 31 + force)
    for m, n in ((5000, 600), (6000, 30), (3000000, 96)):      # (30 columns: 16 lanes x 8 keys per column for runs of 65 .. 128)
        cnt = per_col * n * 2 // 3
        A = esp.ExtendableSparseMatrix(m, n)
        A.debug_force_path(force)
        O = orc.ExtendableSparseMatrix(m, n)
        for rnd in range(2):
            kinds = rng.choice(np.array([0, 1, 1, 1, 2], np.uint8), cnt)
            # two thirds of the columns, per_col entries each at most (exactly per_col in a few)
            cols = rng.permutation(n)[: 2 * n // 3] + 1
            J = rng.permutation(np.repeat(cols, per_col))[:cnt]
            J[: per_col] = cols[0]
            span = 200 if m == 5000 else m
            I = np.minimum(m, np.maximum(1, (J * (m // n)) + rng.integers(-span, span, cnt)))
            dup = rng.random(cnt) < 0.5
            I[dup] = np.minimum(m, J[dup] * (m // n) + rng.integers(0, 5, dup.sum()))
            V = np.where(rng.random(cnt) < 0.1, 0.0, rng.standard_normal(cnt))
            A.append(0, I, J, V, kinds=kinds)
            O.apply(kinds, I, J, V)
            A.flush()
            O.flush()
            assert A.debug_last_path() == 1
            assert_csc_equal(hip_arrays(A), O.arrays(), "m %d round %d" % (m, rnd))


def test_item_producer_shuffled_fem(esp, orc):
    """A shuffled FEM stream on an empty buffer: the producer partitions its ITEMS (a cell's updates for one vertex
    column) and stores every update once, at its bucket position (femitems.hpp; the flush starts at the bucket kernel:
    esp_debug_last_partition 4).  Against the oracle fed the same stream: by itself, with the producer in stream order
    (force_path 25: the flush's own passes), with further appends behind the batch, twice in a row, over a stored CSC,
    after reset!, and with packed instead of 4-byte keys (force_path 14)."""
    rng = np.random.default_rng(8)
    for dim, npd in ((2, 300), (3, 31), (3, 18)):
        nn = npd ** dim
        I, J, V = orc.fem_stream(dim, npd, seed=0x5EED0004, order_mode=1)
        O = orc.ExtendableSparseMatrix(nn, nn)
        O.apply(np.full(len(I), RAW, np.uint8), I, J, V)
        want = O.arrays()
        A = esp.ExtendableSparseMatrix(nn, nn)
        A.generate_fem(dim, npd, seed=0x5EED0004, order_mode=1)
        A.flush()
        assert A.debug_last_partition() == 4 and A.debug_last_key_bytes() == 4, A.debug_last_partition()
        assert_csc_equal(hip_arrays(A), want, "fem %d-D %d" % (dim, npd))
        # (long column runs on a fresh matrix, 4-byte keys: the group tier's kernel with three workgroups per CU, group3.hpp;
        # force_path 30: local_k's group-tier kernels)
        assert A.debug_last_local_small() == 2, (dim, npd, A.debug_last_local_small())
        G = esp.ExtendableSparseMatrix(nn, nn)
        G.debug_force_path(30)
        G.generate_fem(dim, npd, seed=0x5EED0004, order_mode=1)
        G.flush()
        assert G.debug_last_partition() == 4 and G.debug_last_local_small() != 2
        assert_csc_equal(hip_arrays(G), want, "no group3")
        C = esp.ExtendableSparseMatrix(nn, nn)
        C.debug_force_path(14)
        C.generate_fem(dim, npd, seed=0x5EED0004, order_mode=1)
        C.flush()
        assert C.debug_last_partition() == 4 and C.debug_last_key_bytes() == 8
        assert_csc_equal(hip_arrays(C), want, "packed keys")
        L3 = esp.ExtendableSparseMatrix(nn, nn)
        L3.debug_force_path(32)       # the passes stop early, the expansion orders every segment by the last bits (segexpand.hpp)
        L3.generate_fem(dim, npd, seed=0x5EED0004, order_mode=1)
        L3.flush()
        assert L3.debug_last_partition() == 4
        assert_csc_equal(hip_arrays(L3), want, "local bits")
        D = esp.ExtendableSparseMatrix(nn, nn)
        D.debug_force_path(28)        # 16-byte item records (the cell's number in a second word)
        D.generate_fem(dim, npd, seed=0x5EED0004, order_mode=1)
        D.flush()
        assert D.debug_last_partition() == 4
        assert_csc_equal(hip_arrays(D), want, "two-word items")
        B = esp.ExtendableSparseMatrix(nn, nn)
        B.debug_force_path(25)
        B.generate_fem(dim, npd, seed=0x5EED0004, order_mode=1)
        B.flush()
        assert B.debug_last_partition() in (1, 2)
        assert_csc_equal(hip_arrays(B), want, "stream order")
        # re-assembly over the stored pattern (all hits), then appends behind the buckets, then the stream twice
        cnt = 3000
        I2 = rng.integers(1, nn + 1, cnt)
        J2 = rng.integers(1, nn + 1, cnt)
        V2 = rng.standard_normal(cnt)
        k2 = rng.choice(np.array([0, 1, 2], np.uint8), cnt)
        A.generate_fem(dim, npd, seed=0x5EED0004, order_mode=1)
        A.append(0, I2, J2, V2, kinds=k2)
        A.flush()
        O.apply(np.full(len(I), RAW, np.uint8), I, J, V)
        O.apply(k2, I2, J2, V2)
        O.flush()
        assert_csc_equal(hip_arrays(A), O.arrays(), "stored + tail")
        A.reset()
        O.reset()
        A.generate_fem(dim, npd, seed=0x5EED0004, order_mode=1)
        A.generate_fem(dim, npd, seed=0x5EED0005, order_mode=1)
        A.flush()
        Ib, Jb, Vb = orc.fem_stream(dim, npd, seed=0x5EED0005, order_mode=1)
        O.apply(np.full(len(I), RAW, np.uint8), I, J, V)
        O.apply(np.full(len(Ib), RAW, np.uint8), Ib, Jb, Vb)
        O.flush()
        assert_csc_equal(hip_arrays(A), O.arrays(), "twice")


def test_append_elements(esp, orc, monkeypatch):
    """esp_append_elements[_host]: the loops of test/femtools.jl:61-69 for element data held in arrays, against the oracle
    fed the same calls one by one.  On an empty buffer the library partitions (cell, local column) items and the flush starts
    at the bucket kernel (esp_debug_last_partition 4); meshes with natural AND permuted node numbering (nothing may lean on
    grid arithmetic), host and device arrays, without a diagonal term, as updateindex! with zeros, with `-`, a cell that names a
    node twice (stream-order fall-back), behind other appends, twice in a row, over a stored pattern, BoundsError."""
    import torch
    rng = np.random.default_rng(21)
    for dim, npd in ((2, 300), (3, 31), (3, 18)):
        nn = npd ** dim
        nloc = dim + 1
        for node_mode in (0, 1):
            cn, em, dg = orc.fem_mesh(dim, npd, seed=0x5EED0004, order_mode=1, node_mode=node_mode)
            nc = cn.shape[1]
            I, J, V = orc.elements_stream(cn, em, dg)
            O = orc.ExtendableSparseMatrix(nn, nn)
            O.apply(np.full(len(I), RAW, np.uint8), I, J, V)
            want = O.arrays()
            if node_mode == 0:   # (natural numbering: the stream IS generate_fem's)
                I0, J0, V0 = orc.fem_stream(dim, npd, seed=0x5EED0004, order_mode=1)
                assert np.array_equal(I, I0) and np.array_equal(J, J0) and np.array_equal(bits(V), bits(V0))
            # host arrays
            A = esp.ExtendableSparseMatrix(nn, nn)
            A.append_elements(cn, em, dg)
            assert A.nnznew() == len(I)
            A.flush()
            assert A.debug_last_partition() == 4 and A.debug_last_key_bytes() == 4, (A.debug_last_partition(), A.debug_last_key_bytes())
            assert_csc_equal(hip_arrays(A), want, "elements host %d-D %d nodes %d" % (dim, npd, node_mode))
            # device arrays made by esp_generate_fem_mesh: bit for bit the oracle's mesh
            B = esp.ExtendableSparseMatrix(nn, nn)
            dcn = torch.empty((nc, nloc), dtype=torch.int64, device="cuda")
            dem = torch.empty((nc, nloc, nloc), dtype=torch.float64, device="cuda")
            ddg = torch.empty((nc, nloc), dtype=torch.float64, device="cuda")
            B.generate_fem_mesh(dim, npd, dcn, dem, ddg, seed=0x5EED0004, order_mode=1, node_mode=node_mode)
            B.synchronize()
            assert np.array_equal(dcn.cpu().numpy().T, cn)
            assert np.array_equal(bits(dem.cpu().numpy().transpose(2, 1, 0)), bits(em))
            assert np.array_equal(bits(ddg.cpu().numpy().T), bits(dg))
            B.append_elements(dcn, dem, ddg)
            B.flush()
            assert B.debug_last_partition() == 4
            assert_csc_equal(hip_arrays(B), want, "elements device")
            # the kept plan (esp_elements_keep_plan / esp_append_elements_again): the same connectivity with new element data
            # -- a time step -- over the stored pattern, then on the matrix after reset!
            P = esp.ExtendableSparseMatrix(nn, nn)
            with pytest.raises(esp.EspError):
                P.append_elements_again(dem, ddg)                 # (no plan yet)
            P.elements_keep_plan()
            P.append_elements(dcn, dem, ddg)
            P.flush()
            assert_csc_equal(hip_arrays(P), want, "planned first assembly")
            OP = orc.ExtendableSparseMatrix(nn, nn)
            OP.apply(np.full(len(I), RAW, np.uint8), I, J, V)
            for step, (fe, fd) in enumerate(((1.7, 0.3), (-0.25, 2.0))):
                dem2, ddg2 = (dem * fe).contiguous(), (ddg * fd).contiguous()
                Is, Js, Vs = orc.elements_stream(cn, np.asfortranarray(em * fe), np.asfortranarray(dg * fd))
                if step == 1:
                    P.reset()
                    OP.reset()
                if step == 0:
                    P.append_elements_again(dem2, ddg2)
                else:                                             # (host arrays)
                    P.append_elements_again(np.asfortranarray(em * fe), np.asfortranarray(dg * fd))
                assert P.nnznew() == len(I)
                P.flush()
                assert P.debug_last_partition() == 4
                OP.apply(np.full(len(Is), RAW, np.uint8), Is, Js, Vs)
                OP.flush()
                assert_csc_equal(hip_arrays(P), OP.arrays(), "planned step %d" % step)
            with pytest.raises(esp.EspError):
                P.append_elements_again(dem, None)                # (the planned call had a diagonal term)
            # the plan is released while a batch of append_elements_again is still PENDING as a list of items (it reads the plan's
            # item order and cell records at flush time: esp_elements_keep_plan(0) forms its updates first); the tensors the batch
            # was made from are temporaries that go out of scope before the flush (the handle keeps them alive: _Handle.hold)
            P.append_elements_again((dem * 3.0).contiguous(), (ddg * 0.5).contiguous())
            P.elements_keep_plan(False)
            filler = torch.full((dem.numel() + ddg.numel(),), float("nan"), dtype=torch.float64, device="cuda")   # (what a freed tensor's memory would be reused for)
            P.flush()
            del filler
            Is, Js, Vs = orc.elements_stream(cn, np.asfortranarray(em * 3.0), np.asfortranarray(dg * 0.5))
            OP.apply(np.full(len(Is), RAW, np.uint8), Is, Js, Vs)
            OP.flush()
            assert_csc_equal(hip_arrays(P), OP.arrays(), "plan released under a pending batch")
            with pytest.raises(esp.EspError):
                P.append_elements_again(dem, ddg)
            # packed keys, the item partition off (stream order through the flush's own passes), and without the cell
            # records (the expansion gathers from the caller's arrays, as it does for other cell sizes)
            for force, parts in ((14, (4,)), (25, (1, 2)), (37, (4,)), (32, (4,))):   # (37: no cell records; 32: the expansion resolves the last bits)
                Cc = esp.ExtendableSparseMatrix(nn, nn)
                Cc.debug_force_path(force)
                Cc.append_elements(dcn, dem, ddg)
                Cc.flush()
                assert Cc.debug_last_partition() in parts, (force, Cc.debug_last_partition())
                assert_csc_equal(hip_arrays(Cc), want, "force %d" % force)
            # re-assembly over the stored pattern (all hits), then twice in a row (the second call goes behind the batch)
            A.append_elements(dcn, dem, ddg)
            A.flush()
            # (additions over the pattern the same mesh built: the group kernel's re-assembly form -- 4, or 5 in its wide form)
            assert A.debug_last_local_small() in (4, 5), A.debug_last_local_small()
            O.apply(np.full(len(I), RAW, np.uint8), I, J, V)
            O.flush()
            assert_csc_equal(hip_arrays(A), O.arrays(), "re-assembly")
            # ... a third time with other values, as updateindex! calls; then with the re-assembly kernel off (force_path 34)
            em3 = (dem * 0.37).contiguous()
            A.append_elements(dcn, em3, ddg, kind=UPDATE)
            A.flush()
            assert A.debug_last_local_small() in (4, 5)
            I3, J3, V3 = orc.elements_stream(cn, np.asfortranarray(em * 0.37), dg)
            O.apply(np.full(len(I3), UPDATE, np.uint8), I3, J3, V3)
            O.flush()
            assert_csc_equal(hip_arrays(A), O.arrays(), "third assembly")
            A.debug_force_path(34)
            A.append_elements(dcn, dem, ddg)
            A.flush()
            assert A.debug_last_local_small() not in (4, 5)
            A.debug_force_path(0)
            O.apply(np.full(len(I), RAW, np.uint8), I, J, V)
            O.flush()
            assert_csc_equal(hip_arrays(A), O.arrays(), "re-assembly without the re-assembly kernel")
            # a batch that is NOT a re-assembly of the pattern (one cell's column gets a coupling it never had): all-or-nothing
            cnx = cn.copy(order="F")
            cnx[0, 0] = cn[0, nc // 2]
            if len(set(cnx[:, 0].tolist())) == nloc:
                A.append_elements(cnx, em, dg)
                A.flush()
                Ix, Jx, Vx = orc.elements_stream(cnx, em, dg)
                O.apply(np.full(len(Ix), RAW, np.uint8), Ix, Jx, Vx)
                O.flush()
                assert_csc_equal(hip_arrays(A), O.arrays(), "a batch with new couplings over the stored pattern")
            A.reset()
            O.reset()
            A.append_elements(dcn, dem, ddg)
            A.append_elements(cn, em, None, op="-")
            A.flush()
            I2, J2, V2 = orc.elements_stream(cn, em, None)
            O.apply(np.full(len(I), RAW, np.uint8), I, J, V)
            O.apply(np.full(len(I2), RAW, np.uint8), I2, J2, -V2)
            O.flush()
            assert_csc_equal(hip_arrays(A), O.arrays(), "twice, the second without diag and with -")
        # updateindex! with zeros (absent & zero -> no entry), no diagonal term
        em0 = em.copy(order="F")
        em0[rng.random(em0.shape) < 0.3] = 0.0
        I, J, V = orc.elements_stream(cn, em0, None)
        O = orc.ExtendableSparseMatrix(nn, nn)
        O.apply(np.full(len(I), UPDATE, np.uint8), I, J, V)
        A = esp.ExtendableSparseMatrix(nn, nn)
        A.append_elements(cn, em0, None, kind=UPDATE)
        A.flush()
        assert A.debug_last_partition() == 4
        assert_csc_equal(hip_arrays(A), O.arrays(), "UPDATE with zeros")
        # behind other appends (non-empty buffer: stream order), and a cell that names a node twice
        cnt = 3000
        Ia, Ja, Va = rng.integers(1, nn + 1, cnt), rng.integers(1, nn + 1, cnt), rng.standard_normal(cnt)
        ka = rng.choice(np.array([0, 1, 2], np.uint8), cnt)
        cnd = cn.copy(order="F")
        for c in rng.integers(0, nc, 5):
            cnd[1, c] = cnd[0, c]
        I, J, V = orc.elements_stream(cnd, em, dg)
        for first in (True, False):
            O = orc.ExtendableSparseMatrix(nn, nn)
            A = esp.ExtendableSparseMatrix(nn, nn)
            if first:
                O.apply(ka, Ia, Ja, Va)
                A.append(0, Ia, Ja, Va, kinds=ka)
            O.apply(np.full(len(I), RAW, np.uint8), I, J, V)
            A.append_elements(cnd, em, dg)
            A.flush()
            assert A.debug_last_partition() in (1, 2)
            assert_csc_equal(hip_arrays(A), O.arrays(), "repeated node, behind appends: %s" % first)
        # BoundsError: nothing is appended
        cnb = cn.copy(order="F")
        cnb[nloc - 1, nc // 2] = nn + 1
        A = esp.ExtendableSparseMatrix(nn, nn)
        with pytest.raises(esp.BoundsError):
            A.append_elements(cnb, em, dg)
        assert A.nnznew() == 0
        A.append(RAW, Ia, Ja, Va)
        with pytest.raises(esp.BoundsError):
            A.append_elements(cnb, em, dg)
        assert A.nnznew() == cnt


@pytest.mark.parametrize("nloc", [1, 2, 6, 10, 16])
def test_append_elements_any_nloc(esp, orc, nloc):
    """Cells of 1 .. 16 nodes (P2 triangles: 6, P2 tetrahedra: 10): random cells of distinct nodes out of a neighbourhood,
    random element matrices, with and without the diagonal term, rectangular matrices too."""
    rng = np.random.default_rng(100 + nloc)
    for (m, n, nc) in ((40000, 40000, 30000), (50000, 30011, 20000)):
        lim = min(m, n)
        start = rng.integers(0, lim - 64, nc)
        cn = np.empty((nloc, nc), np.int64, order="F")
        for c in range(nc):
            cn[:, c] = 1 + start[c] + rng.choice(64, nloc, replace=False)
        em = np.asfortranarray(rng.standard_normal((nloc, nloc, nc)))
        dg = np.asfortranarray(rng.standard_normal((nloc, nc)))
        for diag in (dg, None):
            I, J, V = orc.elements_stream(cn, em, diag)
            O = orc.ExtendableSparseMatrix(m, n)
            O.apply(np.full(len(I), RAW, np.uint8), I, J, V)
            A = esp.ExtendableSparseMatrix(m, n)
            A.append_elements(cn, em, diag)
            A.flush()
            if nloc > 1 or diag is not None:
                assert A.debug_last_partition() == 4, (nloc, A.debug_last_partition())
            assert_csc_equal(hip_arrays(A), O.arrays(), "nloc %d %dx%d" % (nloc, m, n))


def test_general_path_fdrand_and_plus_mode(esp, orc):
    A = esp.ExtendableSparseMatrix(20 ** 3, 20 ** 3)
    A.debug_force_path(2)
    esp.fdrand_device_(A, 20, 20, 20, rand_mode=1, seed=77)
    O = orc.fdrand(20, 20, 20, rand_mode=1, seed=77, style=orc.KIND_UPDATE)
    assert A.debug_last_path() == 2
    assert_csc_equal(hip_arrays(A), O.arrays())


def test_empty_and_tiny(esp):
    A = esp.ExtendableSparseMatrix(1, 1)
    A.flush()
    assert A.nnz() == 0 and np.array_equal(A.getcolptr(), [1, 1])
    A.updateindex("+", 3.0, 1, 1)
    assert A.nnz() == 1 and A[1, 1] == 3.0
    B = esp.ExtendableSparseMatrix(3, 100000)      # many empty columns
    B[2, 77777] = 1.0
    B[3, 5] = 2.0
    cp, rv, nz = hip_arrays(B)
    check_julia_invariants(3, 100000, cp, rv, nz)
    assert list(rv) == [3, 2] and cp[5] == 2 and cp[77777] == 3 and cp[-1] == 3


# ------------------------------------------------------------------ full-size properties
@pytest.mark.parametrize("force", [0, 5, 12, 14, 15, 16])
def test_run_partition_vs_passes(esp, orc, force):
    """Pre-sorted stream (48^3 stencil, E > 2^20): the producer-side partition (the generator writes every update
    straight to its bucket), the run-based single-pass partition of the flush and the 8-bit passes give the same
    bits; a shuffled stream falls back to the passes."""
    n = 48
    N = n ** 3
    A = esp.ExtendableSparseMatrix(N, N)
    A.debug_force_path(force)
    A.generate_fdrand(n, n, n, seed=21, rand_mode=1)
    A.flush()
    # 4: the producer's own partition (0; 14: with packed keys, 15: generic fold); 16: never the producer's ->
    # 1: histogram + scatter kernel of the flush; 12: its radix-ordered run list; 5: 8-bit passes
    assert A.debug_last_partition() == {0: 4, 5: 2, 12: 1, 14: 4, 15: 4, 16: 1}[force]
    if force in (16, 12):  # run offsets from the ranking kernel / from the radix-ordered run list
        assert A.debug_last_run_order() == (1 if force == 16 else 2)
    # one kind for the whole batch: the bucket kernel reads 4-byte keys (14: packed keys;
    # 0: the UPDATE-only variant of the register tiers, 15: the generic fold on 4-byte keys)
    assert A.debug_last_key_bytes() == (4 if force in (0, 15, 16) else 8)
    O = orc.fdrand(n, n, n, rand_mode=1, seed=21, style=orc.KIND_UPDATE)
    assert_csc_equal(hip_arrays(A), O.arrays())
    # same entries in random order: too many distinct digits per tile -> 8-bit passes
    I, J, V = orc.fdrand_stream(n, n, n, rand_mode=1, seed=22)
    perm = np.random.default_rng(5).permutation(len(I))
    B = esp.ExtendableSparseMatrix(N, N)
    B.debug_force_path(force)
    B.append(UPDATE, I[perm], J[perm], V[perm])
    B.flush()
    assert B.debug_last_partition() == 2
    Ob = orc.ExtendableSparseMatrix(N, N)
    Ob.apply(np.full(len(I), UPDATE, np.uint8), I[perm], J[perm], V[perm])
    assert_csc_equal(hip_arrays(B), Ob.arrays())
    # re-assembly on the existing pattern through the same partition
    A.generate_fdrand(n, n, n, seed=23, rand_mode=1)
    A.flush()
    O.fdrand(n, n, n, rand_mode=1, seed=23, style=orc.KIND_UPDATE)
    # (fdrand! zeroes first; the device matrix accumulated on top: compare against the same sequence)
    O2 = orc.fdrand(n, n, n, rand_mode=1, seed=21, style=orc.KIND_UPDATE)
    I3, J3, V3 = orc.fdrand_stream(n, n, n, rand_mode=1, seed=23)
    O2.apply(np.full(len(I3), UPDATE, np.uint8), I3, J3, V3)
    assert_csc_equal(hip_arrays(A), O2.arrays())


@pytest.mark.parametrize("force", [0, 5, 12])
def test_presorted_stream_mixed_kinds(esp, orc, force):
    """Pre-sorted stream with SET/UPDATE/RAWUPDATE mixes, zeros and duplicates spread over several chunks:
    the run-based single-pass partition (0) and the 8-bit passes (5) give the oracle's bits, on a fresh
    matrix, on re-assembly over the existing pattern and with new positions among hits."""
    rng = np.random.default_rng(77)
    m, n = 150000, 120000
    A = esp.ExtendableSparseMatrix(m, n)
    A.debug_force_path(force)
    O = orc.ExtendableSparseMatrix(m, n)
    for rnd in range(3):
        cnt = 1500000
        # columns drift upwards with jitter (assembly-loop locality); three interleaved sub-streams
        base = np.sort(rng.integers(1, n + 1, cnt))
        off = np.where(np.arange(cnt) % 3 == 1, 3000, 0) + np.where(np.arange(cnt) % 3 == 2, -2500, 0)
        J = (base + off + rng.integers(-40, 41, cnt) - 1) % n + 1
        I = np.clip(J + rng.integers(-3, 4, cnt) * (1 + rnd), 1, m)
        kinds = rng.integers(0, 3, cnt).astype(np.uint8)
        V = np.where(rng.random(cnt) < 0.15, 0.0, rng.standard_normal(cnt))
        A.append(0, I, J, V, kinds=kinds)
        O.apply(kinds, I, J, V)
        A.flush()
        O.flush()
        assert A.debug_last_path() == 1
        assert A.debug_last_partition() == (2 if force == 5 else 1)
        assert A.debug_last_key_bytes() == 8          # per-entry kinds: packed keys
        assert_csc_equal(hip_arrays(A), O.arrays(), "round %d" % rnd)


def test_four_byte_keys_only_for_a_batch_of_one_kind(esp, orc):
    """The bucket kernel gets 4-byte keys only when EVERY pending entry was appended with one known kind; batches of
    several appends, a different kind, a per-entry kinds array, packed keys from outside, a flush in between (the
    bookkeeping starts over), re-assembly over the stored pattern: the oracle's bits each time."""
    rng = np.random.default_rng(123)
    m, n = 90000, 110000

    def stream(cnt):
        J = np.sort(rng.integers(1, n + 1, cnt))
        I = np.clip(J + rng.integers(-4, 5, cnt), 1, m)
        return I, J, rng.standard_normal(cnt)

    A = esp.ExtendableSparseMatrix(m, n)
    O = orc.ExtendableSparseMatrix(m, n)
    plans = [
        ([UPDATE, UPDATE], None, 4),          # two appends, one kind
        ([RAW, RAW, RAW], None, 4),           # re-assembly + new entries, another kind than the last batch
        ([UPDATE, SET], None, 8),             # two kinds
        ([SET], None, 4),
        ([UPDATE], "kinds", 8),               # per-entry kinds array (all equal, but unknown to the host)
        ([UPDATE], "packed", 8),              # packed keys handed over on the device
        ([UPDATE], None, 4),
    ]
    for rnd, (kinds_of, special, expect) in enumerate(plans):
        for kd in kinds_of:
            I, J, V = stream(1000000)
            if special == "kinds":
                A.append(0, I, J, V, kinds=np.full(len(I), kd, np.uint8))
            elif special == "packed":
                import ctypes
                src = sharded_model.HipShardBackend(m, n)
                src.A.append(kd, I, J, V)
                kk, vv, _ = src.shard_export(1)             # packed keys + values on the device, append order
                d = A._d
                d.commit()
                d.ck(d.lib.esp_append_packed(d.h, ctypes.c_void_p(kk.data_ptr()), ctypes.c_void_p(vv.data_ptr()), kk.numel()))
                d.ck(d.lib.esp_synchronize(d.h))
                A._touch()
            else:
                A.append(kd, I, J, V)
            O.apply(np.full(len(I), kd, np.uint8), I, J, V)
        A.flush()
        O.flush()
        assert A.debug_last_partition() == 1, rnd
        assert A.debug_last_key_bytes() == expect, (rnd, A.debug_last_key_bytes())
        assert_csc_equal(hip_arrays(A), O.arrays(), "round %d" % rnd)


@pytest.mark.parametrize("force", [0, 13])
@pytest.mark.parametrize("n", [300007, 2048 * 64, 1000])
def test_colptr_written_by_the_bucket_kernel(esp, orc, n, force):
    """Fresh matrix: every segment writes the colptr of its own columns (0) / column-end marks + scan (13).
    Empty columns in front, between and behind the entries, a column count that is no multiple of the segment
    width, whole empty segments, and a second flush (merge path) on top: the oracle's arrays each time."""
    rng = np.random.default_rng(5 + n)
    m = 50000
    cnt = 1400000 if n > 1000 else 30000
    used = np.sort(rng.choice(np.arange(n // 10, n - n // 7), size=max(1, n // 3), replace=False)) + 1   # 2/3 of the columns stay empty
    J = np.sort(used[rng.integers(0, len(used), cnt)])
    J = J[~((J > n // 2) & (J < n // 2 + n // 8))]                      # a hole of n/8 columns (whole empty segments)
    cnt = len(J)
    I = rng.integers(1, m + 1, cnt)
    V = rng.standard_normal(cnt)
    kinds = rng.integers(0, 3, cnt).astype(np.uint8)
    A = esp.ExtendableSparseMatrix(m, n)
    A.debug_force_path(force)
    O = orc.ExtendableSparseMatrix(m, n)
    A.append(0, I, J, V, kinds=kinds)
    O.apply(kinds, I, J, V)
    A.flush()
    O.flush()
    assert A.debug_last_path() == 1
    assert A.debug_last_colptr_direct() == (force == 0)
    assert_csc_equal(hip_arrays(A), O.arrays(), "fresh")
    J2 = np.sort(rng.integers(1, n + 1, 20000))
    I2 = rng.integers(1, m + 1, 20000)
    V2 = rng.standard_normal(20000)
    A.append(UPDATE, I2, J2, V2)
    O.apply(np.full(20000, UPDATE, np.uint8), I2, J2, V2)
    A.flush()
    O.flush()
    assert not A.debug_last_colptr_direct()
    assert_csc_equal(hip_arrays(A), O.arrays(), "second flush")


def test_digit_with_many_runs_falls_back_to_ordered_run_list(esp, orc):
    """A pre-sorted stream in which every chunk also touches the first columns: that digit collects one run per
    chunk, more than its list holds -> the ranking kernel gives up, nothing is moved, the radix-ordered run list
    takes over; same bits as the oracle.  Without those entries the ranking kernel serves the stream."""
    rng = np.random.default_rng(91)
    m = n = 200000
    cnt = 1500000
    J = np.sort(rng.integers(1, n + 1, cnt))
    I = np.clip(J + rng.integers(-3, 4, cnt), 1, m)
    V = rng.standard_normal(cnt)
    for hot in (True, False):
        Jh = J.copy()
        Ih = I.copy()
        if hot:
            Jh[::1000] = 1 + (np.arange(len(Jh[::1000])) % 3)   # one entry per chunk (4096 entries) and more
            Ih[::1000] = 1 + (np.arange(len(Ih[::1000])) % 5)
        kinds = rng.integers(0, 3, cnt).astype(np.uint8)
        A = esp.ExtendableSparseMatrix(m, n)
        O = orc.ExtendableSparseMatrix(m, n)
        A.append(0, Ih, Jh, V, kinds=kinds)
        O.apply(kinds, Ih, Jh, V)
        A.flush()
        O.flush()
        assert A.debug_last_partition() == 1
        assert A.debug_last_run_order() == (3 if hot else 1)
        assert_csc_equal(hip_arrays(A), O.arrays(), "hot %s" % hot)


def test_producer_side_partition_mixed_with_other_appends(esp, orc):
    """The append is the partition only for a producer that finds the buffer empty; whatever follows (a second
    generator call, a host append, a clone, a changed column window) first turns the bucket-ordered batch back into
    an ordinary pending buffer (4-byte keys expanded): all must stay exact."""
    n = 48
    N = n ** 3
    half = (N // 2 // 256) * 256 + 17
    I, J, V = orc.fdrand_stream(n, n, n, rand_mode=1, seed=31)
    for variant in ("alone", "two_calls", "host_append_after", "host_append_between", "clone", "window", "reassembly",
                    "clear_then_again"):
        A = esp.ExtendableSparseMatrix(N, N)
        O = orc.ExtendableSparseMatrix(N, N)
        upd = lambda I, J, V: O.apply(np.full(len(I), UPDATE, np.uint8), I, J, V)
        expect = 4
        if variant == "alone":
            A.generate_fdrand(n, n, n, seed=31, rand_mode=1)
            upd(I, J, V)
        elif variant in ("two_calls", "host_append_between"):
            A.generate_fdrand_range(n, n, n, 0, half, seed=31, rand_mode=1)
            if variant == "host_append_between":
                A.append(UPDATE, [3, 4], [5, 6], [1.5, 2.5])
            A.generate_fdrand_range(n, n, n, half, N, seed=31, rand_mode=1)
            # stream position of node `half`
            Ia, Ja, Va = orc.fdrand_stream(n, n, n, rand_mode=1, seed=31)
            B = esp.ExtendableSparseMatrix(N, N)
            B.debug_force_path(16)
            B.generate_fdrand_range(n, n, n, 0, half, seed=31, rand_mode=1)
            cut = B.nnznew()
            upd(Ia[:cut], Ja[:cut], Va[:cut])
            if variant == "host_append_between":
                upd(np.array([3, 4]), np.array([5, 6]), np.array([1.5, 2.5]))
            upd(Ia[cut:], Ja[cut:], Va[cut:])
            expect = 1
        elif variant == "host_append_after":
            A.generate_fdrand(n, n, n, seed=31, rand_mode=1)
            A.append(UPDATE, [3, 4, 3], [5, 6, 5], [1.5, 2.5, -0.25])
            upd(I, J, V)
            upd(np.array([3, 4, 3]), np.array([5, 6, 5]), np.array([1.5, 2.5, -0.25]))
            expect = 5   # (batch + tail: only the three entries are partitioned at the flush)
        elif variant == "clone":
            A.generate_fdrand(n, n, n, seed=31, rand_mode=1)
            A2 = A.copy()
            A2.flush()
            upd(I, J, V)
            O.flush()
            assert_csc_equal(hip_arrays(A2), O.arrays(), "clone")
            expect = 1   # (the clone read the pending keys: packed again)
        elif variant == "window":
            A.generate_fdrand(n, n, n, seed=31, rand_mode=1)
            A.set_column_window(1, N)    # same window: the partition still serves
            upd(I, J, V)
        elif variant == "reassembly":
            A.generate_fdrand(n, n, n, seed=30, rand_mode=1)
            A.flush()
            assert A.debug_last_partition() == 4
            I0, J0, V0 = orc.fdrand_stream(n, n, n, rand_mode=1, seed=30)
            upd(I0, J0, V0)
            O.flush()
            A.generate_fdrand(n, n, n, seed=31, rand_mode=1)       # hits only, over the stored pattern
            upd(I, J, V)
        elif variant == "clear_then_again":
            A.generate_fdrand(n, n, n, seed=30, rand_mode=1)
            A._d.ck(A._d.lib.esp_clear_pending(A._d.h))
            A._touch()
            A.generate_fdrand(n, n, n, seed=31, rand_mode=1)
            upd(I, J, V)
        A.flush()
        O.flush()
        assert A.debug_last_partition() == expect, variant
        assert_csc_equal(hip_arrays(A), O.arrays(), variant)


def test_csc_plus_buffer_with_batch_and_tail(esp, orc):
    """csc + buffer (ESP_FLUSH_PLUS, Base.:+(lnk, csc): sparsematrixlnk.jl:294-383) folds the buffer by itself and adds
    the result to the stored value ONCE: a producer's batch with entries behind it must go through ONE fold (two pieces),
    never through the split flush of the routed mode -- (csc + (b1 + b2)) + t differs from csc + ((b1 + b2) + t) in the
    last bit."""
    n = 48
    N = n ** 3
    rng = np.random.default_rng(31)
    A0 = esp.ExtendableSparseMatrix(N, N)
    A0.generate_fdrand(n, n, n, seed=30, rand_mode=1)
    A0.flush()
    cp, rv, nz = hip_arrays(A0)
    csc = esp.SparseMatrixCSC(N, N, cp, rv, nz)
    I, J, V = orc.fdrand_stream(n, n, n, rand_mode=1, seed=31)
    pick = np.sort(rng.choice(len(I), 20000, replace=False))        # the tail: positions the batch holds too, and new ones
    It = np.concatenate([I[pick], rng.integers(1, N + 1, 500)])
    Jt = np.concatenate([J[pick], rng.integers(1, N + 1, 500)])
    Vt = rng.standard_normal(len(It))
    x = esp.SparseMatrixHIPCOO(N, N)
    A = esp.ExtendableSparseMatrix(N, N)           # (the device generator appends into the buffer's handle)
    A._d = x._d
    A.generate_fdrand(n, n, n, seed=31, rand_mode=1)
    x.append(UPDATE, It, Jt, Vt)
    got = x + csc
    assert A.debug_last_partition() == 5
    B = orc.ExtendableSparseMatrix(N, N)           # the buffer's own fold: every call lands in the LNK of an empty matrix
    B.apply(np.full(len(I), UPDATE, np.uint8), I, J, V)
    B.apply(np.full(len(It), UPDATE, np.uint8), It, Jt, Vt)
    B.flush()
    bcp, brv, bnz = B.arrays()
    want = orc.SparseMatrixLNK(orc.CSC(N, N, bcp, brv, bnz)) + orc.CSC(N, N, cp, rv, nz)
    assert_csc_equal((got.colptr, got.rowval, got.nzval), want.arrays(), "csc + (batch + tail)")


def test_nine_bit_partition_passes(esp, orc, monkeypatch):
    """Shuffled streams whose plan needs 17 or 18 prefix bits take TWO passes of 9-bit digits instead of three of at most
    8 (espradix::scatter_k<true>: two digits per thread).  The test hook esp_debug_plan_cap makes the plan ask for that
    many bits at a size the oracle can follow; force_path 23 = 8-bit passes only; both must equal the oracle."""
    rng = np.random.default_rng(123)
    m, n, cnt = 3000, 200003, 2500000
    I, J, V = rng.integers(1, m + 1, cnt), rng.integers(1, n + 1, cnt), rng.standard_normal(cnt)
    kinds = rng.choice(np.array([0, 1, 1, 2], np.uint8), cnt)
    O = orc.ExtendableSparseMatrix(m, n)
    O.apply(kinds, I, J, V)
    O.flush()
    want = O.arrays()
    for cap, force in ((24, 0), (12, 0), (24, 23)):
        A = esp.ExtendableSparseMatrix(m, n)
        A.debug_plan_cap(cap)
        A.debug_force_path(force)
        A.append(0, I, J, V, kinds=kinds)
        A.flush()
        assert A.debug_last_partition() == 2 and A.debug_last_path() == 1
        assert_csc_equal(hip_arrays(A), want, "cap %d force %d" % (cap, force))


def test_last_partition_pass_writes_short_keys(esp, orc):
    """A shuffled stream of ONE kind: from the first radix pass that leaves at most 32 key bits below its prefix on, the passes
    write and read 4-byte keys (12 instead of 16 bytes per entry out of a pass, into the next and into the bucket kernel:
    last_key_bytes 4) -- from host arrays (packed on the device) and from device arrays (the first pass at append time, the
    flush resumes), with two passes (here K = 30 bits: both move 4-byte keys) or three (plan_cap 24).  Mixed kinds keep
    packed keys.  A skewed stream -- a third of the entries in 64 columns -- overfills a segment behind the planned passes: a
    further pass over the 4-byte keys.  Onto a stored matrix (hits + new entries) and with SET (last wins) as well, and
    after a flush whose bucket stage failed: all bit-equal to the oracle."""
    import torch
    rng = np.random.default_rng(321)
    m, n, cnt = 3000, 200003, 2500000
    I, J, V = rng.integers(1, m + 1, cnt), rng.integers(1, n + 1, cnt), rng.standard_normal(cnt)
    Js = J.copy()
    skew = rng.random(cnt) < 0.33
    Js[skew] = 1000 + rng.integers(0, 64, int(skew.sum()))
    mixed = rng.choice(np.array([0, 1, 1, 2], np.uint8), cnt)
    for name, kind, cols, kinds, device, cap, want_bytes in (
            ("update host", UPDATE, J, None, False, 0, 4), ("update device", UPDATE, J, None, True, 0, 4),
            ("set host", 0, J, None, False, 0, 4), ("raw device cap 24", 2, J, None, True, 24, 4),
            ("mixed", 0, J, mixed, False, 0, 8), ("skew host", UPDATE, Js, None, False, 0, 4),
            ("skew device", UPDATE, Js, None, True, 0, 4)):
        kk = np.full(cnt, kind, np.uint8) if kinds is None else kinds
        O = orc.ExtendableSparseMatrix(m, n)
        O.apply(kk, I, cols, V)
        O.flush()
        A = esp.ExtendableSparseMatrix(m, n)
        if cap:
            A.debug_plan_cap(cap)
        if device:
            dI, dJ, dV = (torch.as_tensor(x, device="cuda") for x in (I, cols, V))
            torch.cuda.synchronize()
            A.append_device(kind, dI, dJ, dV)
        else:
            A.append(kind, I, cols, V, kinds=kinds)
        if name.startswith(("update", "raw", "skew")):
            # the bucket stage fails once (test hook): the batch stays pending as packed keys -- rebuilt from the partitioned 4-byte
            # keys where two passes moved those -- and the next flush finishes it
            A.debug_fail_next_bucket_stage()
            with pytest.raises(esp.EspError):
                A.flush()
        A.flush()
        assert A.debug_last_partition() == 2 and A.debug_last_path() == 1, name
        assert A.debug_last_key_bytes() == want_bytes, (name, A.debug_last_key_bytes())
        assert_csc_equal(hip_arrays(A), O.arrays(), name)
        if name in ("update host", "update device"):
            # a second shuffled batch onto the stored matrix: hits and new entries
            I2, J2, V2 = rng.integers(1, m + 1, cnt), rng.integers(1, n + 1, cnt), rng.standard_normal(cnt)
            O.apply(np.full(cnt, UPDATE, np.uint8), I2, J2, V2)
            O.flush()
            if device:
                dI, dJ, dV = (torch.as_tensor(x, device="cuda") for x in (I2, J2, V2))
                torch.cuda.synchronize()
                A.append_device(UPDATE, dI, dJ, dV)
            else:
                A.append(UPDATE, I2, J2, V2)
            A.flush()
            assert_csc_equal(hip_arrays(A), O.arrays(), name + " second batch")


def test_short_keys_are_unpacked_for_the_general_path(esp, orc):
    """64 columns, 2^40 rows, 1.28 million updates of one kind on 3 distinct rows (6 667 per position: no prefix brings a
    segment under the bucket kernel's capacity).  The passes go on into the row bits, switch to 4-byte keys once 32 bits are
    left, give up at 2^24 and more segments -- and the general path must find packed keys again (expand_keys_k in a grid-stride
    loop: one workgroup per segment was an invalid launch; found by the parity fuzz, seed 911)."""
    rng = np.random.default_rng(911)
    m, n, cnt = 2 ** 40, 64, 1280000
    rowpool = rng.integers(1, m + 1, 3)
    I, J, V = rowpool[rng.integers(0, 3, cnt)], rng.integers(1, n + 1, cnt), rng.standard_normal(cnt)
    O = orc.ExtendableSparseMatrix(m, n)
    O.apply(np.full(cnt, UPDATE, np.uint8), I, J, V)
    O.flush()
    A = esp.ExtendableSparseMatrix(m, n)
    A.append(UPDATE, I, J, V)
    A.flush()
    assert A.debug_last_path() == 2, A.debug_last_path()
    assert_csc_equal(hip_arrays(A), O.arrays())


def test_producer_batch_with_a_tail(esp, orc):
    """Entries appended BEHIND a producer's bucket-ordered batch leave it as it is: the flush partitions the tail alone
    and the bucket kernel reads every segment as two pieces (last_partition 5).  Kinds of the tail are free (the batch's
    4-byte keys carry one kind); a tail that overfills a segment, or force_path 19, sends the flush back to packed keys and the ordinary
    partition; a tail that is no pre-sorted stream is ordered by 8-bit passes of its own.  Stream order (batch first) decides SET against UPDATE."""
    n = 48
    N = n ** 3
    rng = np.random.default_rng(77)
    I, J, V = orc.fdrand_stream(n, n, n, rand_mode=1, seed=31)
    upd = lambda O, I, J, V: O.apply(np.full(len(I), UPDATE, np.uint8), I, J, V)

    def tail(k, kinds=None, sort=True):
        Jn = rng.integers(1, N + 1, k)
        if sort:
            Jn = np.sort(Jn)
        In = rng.integers(1, N + 1, k)
        Vn = rng.standard_normal(k)
        kn = np.full(k, UPDATE, np.uint8) if kinds is None else rng.choice(np.array(kinds, np.uint8), k)
        return kn, In, Jn, Vn

    for variant in ("fresh_mixed", "two_tails", "stored", "stored_routed_kinds", "stored_one_flush", "stored_packed", "overfull", "unsorted", "force19", "getindex",
                    "fem", "packed", "packed_update"):
        A = esp.ExtendableSparseMatrix(N, N)
        O = orc.ExtendableSparseMatrix(N, N)
        if variant.startswith("packed"):
            A.debug_force_path(14)                         # (the batch holds packed keys: two packed pieces, KEYS 0 / 3)
        expect = 5
        if variant.startswith("stored"):
            A.generate_fdrand(n, n, n, seed=30, rand_mode=1)
            A.flush()
            upd(O, *orc.fdrand_stream(n, n, n, rand_mode=1, seed=30))
            O.flush()
            if variant == "stored_packed":
                A.debug_force_path(14)                     # (the batch of the split flush holds packed keys)
        if variant == "fem":
            m = 400
            Nf = m * m
            A = esp.ExtendableSparseMatrix(Nf, Nf)
            O = orc.ExtendableSparseMatrix(Nf, Nf)
            A.generate_fem(2, m, seed=5, order_mode=0)
            If, Jf, Vf = orc.fem_stream(2, m, seed=5, order_mode=0)
            O.apply(np.full(len(If), RAW, np.uint8), If, Jf, Vf)
            k = 300
            kn, In, Jn, Vn = np.full(k, UPDATE, np.uint8), rng.integers(1, Nf + 1, k), np.sort(rng.integers(1, Nf + 1, k)), rng.standard_normal(k)
            A.append(0, In, Jn, Vn, kinds=kn)
            O.apply(kn, In, Jn, Vn)
        else:
            A.generate_fdrand(n, n, n, seed=31, rand_mode=1)
            upd(O, I, J, V)
        if variant in ("fresh_mixed", "packed"):
            kn, In, Jn, Vn = tail(5000, kinds=[UPDATE, orc.KIND_SET, orc.KIND_RAWUPDATE])
            A.append(0, In, Jn, Vn, kinds=kn)
            O.apply(kn, In, Jn, Vn)
        elif variant in ("two_tails", "packed_update"):
            for k in (700, 1):
                kn, In, Jn, Vn = tail(k)
                A.append(UPDATE, In, Jn, Vn)
                O.apply(kn, In, Jn, Vn)
        elif variant in ("stored", "stored_routed_kinds", "stored_one_flush", "stored_packed"):
            kn, In, Jn, Vn = tail(4000, kinds=None if variant != "stored_routed_kinds" else [UPDATE, orc.KIND_SET])
            if variant == "stored_one_flush":
                A.debug_force_path(22)                     # (two pieces in one flush, as on a fresh matrix)
            A.append(0, In, Jn, Vn, kinds=kn)
            O.apply(kn, In, Jn, Vn)
            expect = 5 if variant == "stored_one_flush" else 6   # (over a stored pattern: the batch alone, then the tail)
        elif variant == "overfull":                        # 6000 further entries in the columns of one segment
            k = 6000
            In, Jn, Vn = rng.integers(1, N + 1, k), np.sort(rng.integers(1000, 1004, k)), rng.standard_normal(k)
            A.append(UPDATE, In, Jn, Vn)
            upd(O, In, Jn, Vn)
            expect = None
        elif variant == "unsorted":
            kn, In, Jn, Vn = tail(300000, sort=False)   # (no pre-sorted stream: 8-bit passes over the tail alone)
            A.append(UPDATE, In, Jn, Vn)
            O.apply(kn, In, Jn, Vn)
        elif variant == "force19":
            A.debug_force_path(19)
            kn, In, Jn, Vn = tail(500)
            A.append(UPDATE, In, Jn, Vn)
            O.apply(kn, In, Jn, Vn)
            expect = None
        elif variant == "getindex":
            kn, In, Jn, Vn = tail(500)
            A.append(UPDATE, In, Jn, Vn)
            O.apply(kn, In, Jn, Vn)
            X = esp.SparseMatrixHIPCOO(N, N)               # (the buffer's API on the same handle)
            X._d = A._d
            i, j = int(In[7]), int(Jn[7])
            want = float(Vn[(In == i) & (Jn == j)].sum())
            assert abs(X[i, j] - want) <= 1e-12 * max(1.0, abs(want))  # (reads the pending entries: packed keys again, tail kept)
            l = int(I[1000]), int(J[1000])
            assert X[l] != 0.0
            expect = None
        A.flush()
        O.flush()
        if expect is not None:
            assert A.debug_last_partition() == expect, variant
        else:
            assert A.debug_last_partition() != 5, variant
        assert_csc_equal(hip_arrays(A), O.arrays(), variant)


def test_producer_side_partition_fem_and_fallbacks(esp, orc):
    """FEM producer in natural cell order: partition by the producer (4-byte keys, RAWUPDATE kind); shuffled cell
    order: the COUNT launch finds too many digits per chunk, the plain producer runs and the flush partitions with
    its 8-bit passes.  A 1-D stencil whose buckets would cut into the row bits does not qualify either."""
    for dim, npd in ((2, 400), (3, 40)):
        nn = npd ** dim
        for order in (0, 1):
            A = esp.ExtendableSparseMatrix(nn, nn)
            A.generate_fem(dim, npd, seed=0x5EED0004, order_mode=order)
            A.flush()
            B = esp.ExtendableSparseMatrix(nn, nn)
            B.debug_force_path(16)
            B.generate_fem(dim, npd, seed=0x5EED0004, order_mode=order)
            B.flush()
            assert A.debug_last_partition() == 4, (dim, order, A.debug_last_partition())   # natural order: run lists; random: item partition
            assert B.debug_last_partition() in (1, 2)
            if order == 0:
                assert A.debug_last_key_bytes() == 4
            assert_csc_equal(hip_arrays(A), hip_arrays(B), "fem %d %d" % (dim, order))
            if order == 0:
                O = orc.ExtendableSparseMatrix(nn, nn)
                I, J, V = orc.fem_stream(dim, npd, seed=0x5EED0004, order_mode=order)
                O.apply(np.full(len(I), RAW, np.uint8), I, J, V)
                O.flush()
                assert_csc_equal(hip_arrays(A), O.arrays(), "fem vs oracle")


@pytest.mark.parametrize("n", [96, 256])
def test_fdrand_large_properties(esp, n):
    """BASELINE config 2 size (256^3) and a mid size: size-independent properties of the result:
    nnz formula, Julia invariants, symmetry of the pattern, zero row sums away from the boundary
    terms (rand=()->1: diagonal = -sum(offdiag) + boundary), and the 7-point pattern."""
    N = n ** 3
    A = esp.ExtendableSparseMatrix(N, N)
    A.generate_fdrand(n, n, n, seed=0x5EED0002, rand_mode=0)
    assert A.nnznew() == 12 * n * n * (n - 1) + 6 * n * n
    A.flush()
    assert A.nnz() == N + 6 * n * n * (n - 1)
    cp, rv, nz = hip_arrays(A)
    check_julia_invariants(N, N, cp, rv, nz)
    cnt = np.diff(cp)
    assert cnt.min() == 4 and cnt.max() == 7
    col = np.repeat(np.arange(1, N + 1, dtype=np.int64), cnt)
    d = rv - col
    ad = np.abs(d)
    assert np.all((ad == 0) | (ad == 1) | (ad == n) | (ad == n * n))
    del ad
    off = nz[d != 0]
    assert np.all(off == -(1.0 / n))                     # -h*h/h
    colsum = np.add.reduceat(nz, cp[:-1] - 1)
    # column sums = boundary terms only: h*h times the number of boundary faces of the node
    g = np.arange(N)
    i, j, k = g % n, (g // n) % n, g // (n * n)
    faces = ((i == 0) | (i == n - 1)).astype(float) + ((j == 0) | (j == n - 1)) + ((k == 0) | (k == n - 1))
    assert np.allclose(colsum, faces / (n * n), rtol=0, atol=1e-12)


# ------------------------------------------------------------------ shards (multi-GPU building blocks)
@pytest.mark.parametrize("P", [1, 2, 3, 8])
def test_shard_export_is_a_stable_partition_by_owner(esp, P):
    import torch
    rng = np.random.default_rng(P)
    m, n = 900, 1000
    cnt = 50000
    be = sharded_model.HipShardBackend(m, n)
    I = rng.integers(1, m + 1, cnt)
    J = rng.integers(1, n + 1, cnt)
    V = rng.standard_normal(cnt)
    kinds = rng.integers(0, 3, cnt).astype(np.uint8)
    be.matrix.append(0, I, J, V, kinds=kinds)
    counts = be.shard_counts(P)
    keys, vals, offsets = be.shard_export(P)
    torch.cuda.synchronize()
    rb = 10  # bits_for(900)
    key = ((((J - 1) << rb) | (I - 1)) << 2) | kinds
    owner = ((J - 1) * P) // n
    order = np.argsort(owner, kind="stable")
    assert np.array_equal(counts, np.bincount(owner, minlength=P))
    assert np.array_equal(offsets, np.concatenate([[0], np.cumsum(counts)]))
    assert np.array_equal(keys.cpu().numpy(), key[order])
    assert np.array_equal(vals.cpu().numpy().view(np.uint64), V[order].view(np.uint64))


def test_sharded_matrix_world1_nccl(esp, orc):
    """The product exchange path end to end on one GPU (world_size 1, RCCL)."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("process group already initialised")
    os_env = __import__("os").environ
    os_env.setdefault("MASTER_ADDR", "127.0.0.1")
    os_env.setdefault("MASTER_PORT", "29531")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        nx, ny, nz = 12, 11, 10
        N = nx * ny * nz
        A = sharded_model.ShardedExtendableSparseMatrix(N, N, sharded_model.HipShardBackend(N, N, device=0))
        A.local.generate_fdrand(nx, ny, nz, seed=3, rand_mode=1)
        A.flush()
        O = orc.fdrand(nx, ny, nz, rand_mode=1, seed=3, style=orc.KIND_UPDATE)
        G = A.gather_sparse(0)
        assert_csc_equal(G.arrays(), O.arrays())
        assert A.nnz() == orc.fdrand_nnz(nx, ny, nz) and A.exchanged == (orc.fdrand_count(nx, ny, nz),) * 2
    finally:
        dist.destroy_process_group()


def test_mul_bitwise_equals_column_loop(esp, orc):
    """mul!(r, ext, x) on the device CSC: bit-identical to the reference's column loop (oracle), on random
    rectangular matrices with empty rows/columns, after numeric re-assembly (same pattern, new values: the
    row-wise index is reused), after a pattern change and after dropzeros!; NumPy and device-tensor forms."""
    import torch
    rng = np.random.default_rng(61)
    for (m, n, cnt) in [(700, 500, 6000), (3, 4000, 5000), (5000, 7, 9000), (40, 40, 0)]:
        A = esp.ExtendableSparseMatrix(m, n)
        O = orc.ExtendableSparseMatrix(m, n)
        for rnd in range(3):
            I = rng.integers(1, m + 1, cnt)
            J = rng.integers(1, n + 1, cnt)
            if rnd == 1:
                I, J = Iprev, Jprev                      # same pattern, other values
            V = rng.standard_normal(cnt) * 10.0 ** rng.integers(-8, 9, cnt)
            Iprev, Jprev = I, J
            A.append(UPDATE, I, J, V)
            O.apply(np.full(cnt, UPDATE, np.uint8), I, J, V)
            x = rng.standard_normal(n) * 10.0 ** rng.integers(-5, 6, n)
            want = O.sparse().mul(x)
            got = A.mul(x)
            assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), (m, n, rnd)
            xt = torch.from_numpy(x).cuda()
            rt = A @ xt
            assert np.array_equal(rt.cpu().numpy().view(np.uint64), want.view(np.uint64))
        A.rawupdateindex("+", 0.0, m, n)                 # a structural zero, then dropped again
        O.rawupdateindex(orc.OP_ADD, 0.0, m, n)
        x = rng.standard_normal(n)
        assert np.array_equal(A.mul(x).view(np.uint64), O.sparse().mul(x).view(np.uint64))
        A.dropzeros()
        O.dropzeros()
        assert np.array_equal(A.mul(x).view(np.uint64), O.sparse().mul(x).view(np.uint64))
    # the stencil: rand=()->1 gives zero row sums away from the boundary terms
    nn = 40
    S = esp.fdrand(nn, nn, nn, rand_mode=0)
    y = S.mul(np.ones(nn ** 3))
    Os = orc.fdrand(nn, nn, nn, rand_mode=0, style=orc.KIND_UPDATE)
    assert np.array_equal(y.view(np.uint64), Os.sparse().mul(np.ones(nn ** 3)).view(np.uint64))
    interior = np.zeros((nn, nn, nn), bool)
    interior[1:-1, 1:-1, 1:-1] = True
    assert np.all(np.abs(y[interior.ravel()]) < 1e-12)
    with pytest.raises(ValueError):
        S.mul(np.ones(5))


def test_dirichlet_helpers(esp, orc):
    """mark_dirichlet / eliminate_dirichlet! (sparsematrixcsc.jl:94-144) on the device CSC == the oracle's loops."""
    rng = np.random.default_rng(71)
    nx, ny, nz = 14, 11, 9
    N = nx * ny * nz
    A = esp.fdrand(nx, ny, nz, rand_mode=1, seed=5)
    O = orc.fdrand(nx, ny, nz, rand_mode=1, seed=5, style=orc.KIND_UPDATE)
    nodes = rng.choice(N, 60, replace=False) + 1
    for i in nodes:                                    # penalty method: A[i,i] += 1e30
        A.updateindex("+", 1.0e30, int(i), int(i))
        O.updateindex(orc.OP_ADD, 1.0e30, int(i), int(i))
    A[5, 7] = 3.0e25                                   # a large off-diagonal entry is not a marker
    O[5, 7] = 3.0e25
    mk = A.mark_dirichlet()
    want = O.sparse().mark_dirichlet()
    assert np.array_equal(mk, want) and mk.sum() == 60 and set(np.flatnonzero(mk) + 1) == set(int(i) for i in nodes)
    A.eliminate_dirichlet(mk)
    C0 = O.sparse()
    C0.eliminate_dirichlet(want)
    assert_csc_equal(hip_arrays(A), C0.arrays())
    x = rng.standard_normal(N)
    assert np.array_equal(A.mul(x).view(np.uint64), C0.mul(x).view(np.uint64))
    with pytest.raises(ValueError):
        A.eliminate_dirichlet(np.zeros(3, bool))


def test_coo_constructor_and_fdrand_coo(esp, orc):
    """ExtendableSparseMatrixCSC(I,J,V[,m,n]) (extendable.jl:85-104) and fdrand_coo (sprand.jl:134-185)
    through the device pipeline as COO entries == the oracle's sparse(I,J,V,m,n,+)."""
    rng = np.random.default_rng(51)
    m, n, cnt = 4000, 3500, 300000
    I = rng.integers(1, m + 1, cnt)
    J = rng.integers(1, n + 1, cnt)
    hot = rng.random(cnt) < 0.25
    I[hot] = rng.integers(1, 60, hot.sum())
    J[hot] = rng.integers(1, 50, hot.sum())
    V = np.where(rng.random(cnt) < 0.2, 0.0, rng.standard_normal(cnt))
    V[rng.random(cnt) < 0.02] = -0.0
    A = esp.ExtendableSparseMatrix.from_coo(I, J, V, m, n)
    assert_csc_equal(hip_arrays(A), orc.sparse_coo(I, J, V, m, n).arrays())
    B = esp.ExtendableSparseMatrix.from_coo(I, J, V)          # sizes from the largest indices
    assert B.shape == (int(I.max()), int(J.max()))
    assert_csc_equal(hip_arrays(B), orc.sparse_coo(I, J, V).arrays())
    with pytest.raises((esp.BoundsError, IndexError)):
        esp.ExtendableSparseMatrix.from_coo([1, 9], [1, 1], [1.0, 2.0], 4, 4)
    # a lone -0.0 keeps its sign; zeros are structural entries
    C0 = esp.ExtendableSparseMatrix.from_coo([2, 3, 3], [4, 1, 1], [-0.0, 1.5, 2.0])
    cp, rv, nz = hip_arrays(C0)
    assert list(cp) == [1, 2, 2, 2, 3] and list(rv) == [3, 2] and nz[0] == 3.5 and np.signbit(nz[1])
    # fdrand_coo: device generator (COO entries) == host triplets == oracle
    nx, ny, nz_ = 9, 8, 7
    N = nx * ny * nz_
    Ist, Jst, Vst = orc.fdrand_stream(nx, ny, nz_, rand_mode=2, seed=6)
    want = orc.sparse_coo(Ist, Jst, Vst, N, N).arrays()
    assert_csc_equal(hip_arrays(esp.fdrand_coo(nx, ny, nz_, rand_mode=2, seed=6)), want)
    assert_csc_equal(hip_arrays(esp.fdrand_coo(nx, ny, nz_, rand_mode=2, seed=6, device=False)), want)
    # large: 96^3 through the run-based partition
    n3 = 96
    D = esp.fdrand_coo(n3, n3, n3, rand_mode=1, seed=12)
    I3, J3, V3 = orc.fdrand_stream(n3, n3, n3, rand_mode=1, seed=12)
    assert_csc_equal(hip_arrays(D), orc.sparse_coo(I3, J3, V3, n3 ** 3, n3 ** 3).arrays())


def test_coo_entries_mixed_with_updates(esp):
    """COO entries among SET/UPDATE/RAWUPDATE, on a fresh matrix and over an existing CSC (ROUTED):
    the dict model (tests/refmodel.py) gives the same bits."""
    from refmodel import COO, DictModel
    rng = np.random.default_rng(52)
    m, n = 300, 200
    A = esp.ExtendableSparseMatrix(m, n)
    M = DictModel(m, n)
    for rnd in range(3):
        cnt = 20000
        I = rng.integers(1, m + 1, cnt)
        J = rng.integers(1, n + 1, cnt)
        V = np.where(rng.random(cnt) < 0.25, 0.0, rng.standard_normal(cnt))
        V[rng.random(cnt) < 0.03] = -0.0
        kinds = rng.integers(0, 4, cnt).astype(np.uint8)
        A.append(0, I, J, V, kinds=kinds)
        for i, j, v, k in zip(I, J, V, kinds):
            M.apply(COO if k == 3 else int(k), v, int(i), int(j))
        A.flush()
        M.flush()
        assert_csc_equal(hip_arrays(A), M.arrays(), "round %d" % rnd)


def _sharded_ranks_run(esp, orc, world, deal, n=44, rounds=2, overflow=False, one_message=True):
    """W ranks as threads on one GPU (tests/threaddist.py): returns per-rank exchange kinds and checks
    the gathered CSC against ONE oracle buffer fed the ranks' streams in rank order."""
    from threaddist import run_ranks
    nx = ny = n
    nzg = n * world
    N = nx * ny * nzg
    nodes = nx * ny * n
    seeds = [77, 78, 79]
    streams = [orc.fdrand_stream(nx, ny, nzg, rand_mode=1, seed=seeds[r]) for r in range(rounds)]
    E = len(streams[0][0])
    kinds = np.where(np.arange(E) % 5 == 0, RAW, UPDATE).astype(np.uint8)
    if deal == "slab":
        # rank r runs the device generator on its z-slab of nodes; in the stream that is a slice:
        # updates per node (sprand.jl:87-126) -> slice boundaries
        l = np.arange(N)
        i, j, k = l % nx + 1, (l // nx) % ny + 1, l // (nx * ny) + 1
        per = (4 * (i < nx) + ((i == 1) | (i == nx)) + 4 * (j < ny) + ((j == 1) | (j == ny))
               + 4 * (k < nzg) + ((k == 1) | (k == nzg)))
        assert per.sum() == E
        off = np.concatenate([[0], np.cumsum(per)])
        sel = [slice(int(off[r * nodes]), int(off[(r + 1) * nodes])) for r in range(world)]
        kinds[:] = UPDATE
    else:
        chunk = np.arange(E) // 4096
        sel = [(chunk % world) == r for r in range(world)]
    perm1 = np.random.default_rng(9).permutation(int(np.count_nonzero(sel[1])) if deal != "slab" else 1)
    extra = None
    if overflow:
        # both ranks pile entries on the same few columns: every rank's own bucket fits the bucket
        # kernel, the merged segment does not
        rng = np.random.default_rng(3)
        extra = (rng.integers(1, N + 1, 2600), rng.integers(N // 2 + 10, N // 2 + 40, 2600), rng.standard_normal(2600))

    def rank_stream(rank, rnd):
        I, J, V = streams[rnd]
        Ii, Jj, Vv, kk = I[sel[rank]], J[sel[rank]], V[sel[rank]], kinds[sel[rank]]
        if deal == "shuffled_rank1" and rank == 1:
            Ii, Jj, Vv, kk = Ii[perm1], Jj[perm1], Vv[perm1], kk[perm1]
        return Ii, Jj, Vv, kk

    def body(rank, dist):
        be = sharded_model.HipShardBackend(N, N, device=0)
        A = sharded_model.ShardedExtendableSparseMatrix(N, N, be, dist=dist)
        hist, folds = [], []
        for rnd in range(rounds):
            if deal == "slab":
                A.local.generate_fdrand_range(nx, ny, nzg, rank * nodes, (rank + 1) * nodes, seed=seeds[rnd], rand_mode=1)
            else:
                Ii, Jj, Vv, kk = rank_stream(rank, rnd)
                A.append(0, Ii, Jj, Vv, kinds=kk)
            if extra is not None and rnd == 0:
                A.append(UPDATE, extra[0], extra[1], extra[2] * (rank + 1))
            A.flush()
            hist.append((A.last_exchange, be.matrix.debug_last_partition()))
            folds.append(be.matrix.debug_last_fold_update())
        G = A.gather_sparse(0)
        return hist, (G.arrays() if rank == 0 else None), A.nnz(), folds

    shmod = sys.modules[sharded_model.ShardedExtendableSparseMatrix.__module__]
    keep = shmod.ONE_MESSAGE_MAX_ELEMS
    if not one_message:
        shmod.ONE_MESSAGE_MAX_ELEMS = 0   # counts, keys, values as three collectives (large exchanges)
    try:
        outs = run_ranks(world, body)
    finally:
        shmod.ONE_MESSAGE_MAX_ELEMS = keep
    # oracle: one buffer, the ranks' streams in rank order, flush after every round
    O = orc.ExtendableSparseMatrix(N, N)
    for rnd in range(rounds):
        for rank in range(world):
            Ii, Jj, Vv, kk = rank_stream(rank, rnd)
            O.apply(kk, Ii, Jj, Vv)
            if extra is not None and rnd == 0:
                O.apply(np.full(len(extra[0]), UPDATE, np.uint8), extra[0], extra[1], extra[2] * (rank + 1))
        O.flush()
    assert_csc_equal(outs[0][1], O.arrays())
    assert outs[0][2] == O.nnz()
    # the UPDATE-only fold of the bucket kernel: a shard uses it when its own batch was appended as UPDATEs AND the
    # device check of the received blocks found nothing else (slab: the generator's updateindex! stream; the
    # other deals come with per-entry kinds, RAWUPDATEs among them)
    for o in outs:
        if o[0] and all(h == ("partitioned", 7) for h in o[0]):
            assert o[3] == [deal == "slab"] * len(o[3]), (deal, o[3])
    return [o[0] for o in outs]


@pytest.mark.parametrize("world,deal", [(2, "slab"), (3, "slab"), (2, "scrambled"), (3, "scrambled")])
def test_partitioned_exchange_ranks_as_threads(esp, orc, world, deal):
    """The partitioned exchange with 2 and 3 source ranks per segment: one partition pass per rank,
    pieces assembled without a copy, bits equal to one buffer fed the ranks' streams in turn; second
    round = re-assembly over the existing CSC (hits applied in place through the pieces)."""
    hist = _sharded_ranks_run(esp, orc, world, deal, one_message=(world == 2))
    for h in hist:
        assert h == [("partitioned", 7), ("partitioned", 7)], h


def test_partitioned_exchange_falls_back_by_consensus(esp, orc):
    """One rank's stream is shuffled: its partition reports "not applicable", ALL ranks take the plain
    exchange for this flush (and back off for the next)."""
    hist = _sharded_ranks_run(esp, orc, 2, "shuffled_rank1")
    for h in hist:
        assert [x[0] for x in h] == ["inplace", "inplace"], h


def test_partitioned_exchange_merged_segment_overflow(esp, orc):
    """Every rank's own bucket fits the bucket kernel but a merged segment does not: esp_shard_assemble
    hands the entries over as a plain pending buffer (rank order) and the flush partitions again."""
    hist = _sharded_ranks_run(esp, orc, 2, "scrambled", rounds=1, overflow=True)
    # the piled-up columns belong to rank 1: its flush partitions again, rank 0 runs on its pieces
    assert hist[0] == [("partitioned", 7)] and hist[1][0][0] == "partitioned" and hist[1][0][1] != 7, hist


def test_failed_fresh_flush_leaves_the_empty_matrix_intact(esp, orc):
    """A fresh flush whose bucket kernel writes colptr itself and then meets an entry outside the declared
    window: the flush fails, colptr is all ones again (nnz 0), and the matrix works afterwards."""
    rng = np.random.default_rng(32)
    m, n = 3000, 40000
    lo, hi = 20001, 21000
    cnt = 3000                                        # one segment: the bucket kernel is the first to look at the keys
    I = rng.integers(1, m + 1, cnt)
    J = rng.integers(lo, hi + 1, cnt)
    V = rng.standard_normal(cnt)
    A = esp.ExtendableSparseMatrix(m, n)
    A.set_column_window(lo, hi)
    A.append(UPDATE, I, J, V)
    A.append(UPDATE, [1, 2, 3], [hi + 5, hi + 6, 7], [1.0, 2.0, 3.0])   # above and BELOW the window (key difference wraps)
    with pytest.raises(esp.EspError):
        A.flush()
    d = A._d                                          # (the accessors would flush again: drop the batch first)
    d.ck(d.lib.esp_clear_pending(d.h))
    A._touch()
    assert A.nnz() == 0
    assert np.all(A.getcolptr() == 1)
    A.reset()
    O = orc.ExtendableSparseMatrix(m, n)
    A.append(UPDATE, I, J, V)
    O.apply(np.full(cnt, UPDATE, np.uint8), I, J, V)
    A.flush()
    assert A.debug_last_colptr_direct()
    assert_csc_equal(hip_arrays(A), O.arrays())


def test_window_violation_below_the_window_with_stored_entries(esp, orc):
    """Entries below a declared window (their key difference wraps around) on a matrix that already holds
    entries: the bucket kernel marks column ends -- inside the array -- and the flush fails cleanly."""
    rng = np.random.default_rng(33)
    m, n = 3000, 40000
    lo, hi = 20001, 21000
    cnt = 3000
    I = rng.integers(1, m + 1, cnt)
    J = rng.integers(lo, hi + 1, cnt)
    V = rng.standard_normal(cnt)
    A = esp.ExtendableSparseMatrix(m, n)
    O = orc.ExtendableSparseMatrix(m, n)
    A.append(UPDATE, I, J, V)
    O.apply(np.full(cnt, UPDATE, np.uint8), I, J, V)
    A.flush()
    A.set_column_window(lo, hi)
    A.append(UPDATE, I[:100], J[:100], V[:100])
    A.append(UPDATE, [5, 6], [3, lo - 1], [1.0, 2.0])
    with pytest.raises(esp.EspError):
        A.flush()
    d = A._d
    d.ck(d.lib.esp_clear_pending(d.h))
    A._touch()
    # the pattern is intact (the values are not specified after a failed flush: updates that hit stored
    # positions are applied in place before the violation is known)
    cp, rv, _ = hip_arrays(A)
    ocp, orv, _ = O.arrays()
    assert np.array_equal(cp, ocp) and np.array_equal(rv, orv)
    A.reset()
    O.reset()
    A.append(UPDATE, I, J, V)
    O.apply(np.full(cnt, UPDATE, np.uint8), I, J, V)
    A.flush()
    assert_csc_equal(hip_arrays(A), O.arrays())


def test_column_window(esp, orc):
    rng = np.random.default_rng(31)
    m, n = 3000, 40000
    lo, hi = 12345, 17000            # 1-based inclusive window, not a power of two
    cnt = 80000
    I = rng.integers(1, m + 1, cnt)
    J = rng.integers(lo, hi + 1, cnt)
    V = rng.standard_normal(cnt)
    A = esp.ExtendableSparseMatrix(m, n)
    A.set_column_window(lo, hi)
    O = orc.ExtendableSparseMatrix(m, n)
    for _ in range(2):
        A.append(UPDATE, I, J, V)
        O.apply(np.full(cnt, UPDATE, np.uint8), I, J, V)
        A.flush()
        assert A.debug_last_path() == 1
        assert_csc_equal(hip_arrays(A), O.arrays())
    A.append(UPDATE, [1] * 5000, [hi + 1] * 5000, [1.0] * 5000)   # outside the window
    with pytest.raises(esp.EspError):
        A.flush()
    # reset! keeps the window: per-column work stays inside it, the rest of colptr is refreshed on demand
    A.reset()
    O.reset()
    for rnd in range(2):
        V2 = rng.standard_normal(cnt)
        A.append(UPDATE, I, J, V2)
        O.apply(np.full(cnt, UPDATE, np.uint8), I, J, V2)
        A.flush()
        assert A[int(I[0]), int(J[0])] == O[int(I[0]), int(J[0])]          # (getindex reads colptr)
        assert_csc_equal(hip_arrays(A), O.arrays())
    # a window declared on a matrix that already holds entries elsewhere only restricts the pending ones
    B = esp.ExtendableSparseMatrix(m, n)
    OB = orc.ExtendableSparseMatrix(m, n)
    Jall = rng.integers(1, n + 1, cnt)
    B.append(UPDATE, I, Jall, V)
    OB.apply(np.full(cnt, UPDATE, np.uint8), I, Jall, V)
    B.flush()
    B.set_column_window(lo, hi)
    B.append(UPDATE, I, J, V)
    OB.apply(np.full(cnt, UPDATE, np.uint8), I, J, V)
    assert_csc_equal(hip_arrays(B), OB.arrays())


def test_generate_fdrand_range_halves(esp, orc):
    nx, ny, nz = 9, 7, 6
    N = nx * ny * nz
    A = esp.ExtendableSparseMatrix(N, N)
    cut = 3 * nx * ny + 17       # inside a plane and inside a row
    A.generate_fdrand_range(nx, ny, nz, 0, cut, seed=8, rand_mode=1)
    A.generate_fdrand_range(nx, ny, nz, cut, N, seed=8, rand_mode=1)
    assert A.nnznew() == orc.fdrand_count(nx, ny, nz)
    A.flush()
    O = orc.fdrand(nx, ny, nz, rand_mode=1, seed=8, style=orc.KIND_UPDATE)
    assert_csc_equal(hip_arrays(A), O.arrays())


def test_all_to_all_large_message(esp):
    """Exchange helper with > 2^27 elements through RCCL (a single all_to_all_single call of that
    size delivers only part of the data on this stack: the helper splits it into rounds)."""
    import torch
    import torch.distributed as dist
    from sharded_model import all_to_all_v
    if dist.is_initialized():
        pytest.skip("process group already initialised")
    os_env = __import__("os").environ
    os_env.setdefault("MASTER_ADDR", "127.0.0.1")
    os_env.setdefault("MASTER_PORT", "29532")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        cnt = 150_000_000
        a = torch.arange(cnt, dtype=torch.int64, device="cuda")
        b = torch.zeros_like(a)
        rounds = all_to_all_v(dist, b, a, [cnt], [cnt])
        torch.cuda.synchronize()
        assert rounds > 1 and torch.equal(a, b)
    finally:
        dist.destroy_process_group()


def test_inplace_exchange_three_shards_on_one_gpu(esp, orc):
    """esp_shard_exchange_begin/place with P=3, emulating the all-to-all between three handles that
    live on the same GPU (device-to-device copies in place of RCCL): every shard must end up with
    exactly the entries of its column range, in (source rank, source order)."""
    import torch
    P = 3
    m, n = 500, 900
    rng = np.random.default_rng(77)
    bes = [sharded_model.HipShardBackend(m, n) for _ in range(P)]
    streams = []
    for r in range(P):
        cnt = 20000 + 3000 * r
        I = rng.integers(1, m + 1, cnt)
        J = rng.integers(1, n + 1, cnt)
        V = rng.standard_normal(cnt)
        K = rng.integers(0, 3, cnt).astype(np.uint8)
        bes[r].matrix.append(0, I, J, V, kinds=K)
        streams.append((K, I, J, V))
    counts = np.stack([be.shard_counts(P) for be in bes])          # counts[src][dst]
    sends = []
    for r in range(P):
        lower = int(sum(counts[s][r] for s in range(r)))
        higher = int(sum(counts[s][r] for s in range(r + 1, P)))
        keys, vals, soff = bes[r].exchange_begin(P, r, lower, higher)
        assert soff[-1] == counts[r].sum() - counts[r][r]
        sends.append((keys, vals, soff, lower))
    torch.cuda.synchronize()
    for dst in range(P):
        pos_lo, pos_hi = 0, sends[dst][3] + int(counts[dst][dst])
        for src in range(P):
            if src == dst:
                continue
            k, v, soff, _ = sends[src]
            ck, cv = k[soff[dst]:soff[dst + 1]].clone(), v[soff[dst]:soff[dst + 1]].clone()
            if src < dst:
                bes[dst].exchange_place(pos_lo, ck, cv)
                pos_lo += ck.numel()
            else:
                bes[dst].exchange_place(pos_hi, ck, cv)
                pos_hi += ck.numel()
    ranges = esp.owner_ranges(n, P)
    for dst in range(P):
        c0, c1 = ranges[dst]
        bes[dst].set_column_window(c0 + 1, c1)
        bes[dst].flush()
        O = orc.ExtendableSparseMatrix(m, n)
        for (K, I, J, V) in streams:                               # rank order, owned columns only
            sel = (J - 1 >= c0) & (J - 1 < c1)
            O.apply(K[sel], I[sel], J[sel], V[sel])
        assert_csc_equal(bes[dst].local_csc().arrays(), O.arrays(), "shard %d" % dst)


# ------------------------------------------------------------------ full-size pins (tests/golden/digests_large.txt)
@pytest.mark.parametrize("n", [128, 192, 256])
def test_fdrand_full_size_digest(esp, n):
    """BASELINE config 2 with random values at the bench's size (256^3: rand_mode 1, seed 0x5EED0002, updateindex!
    style -- exactly 32 key bits below the 16-bit prefix) and two smaller cubes: the device CSC's sha256 equals the
    oracle's (tests/golden/make_digests_large.py).  Producer-side partition and, with the hook, the flush's own."""
    d = gu.digests("digests_large.txt")["fd_%d_m1" % n]
    N = n ** 3
    for force in ((0, 16, 18) if n < 256 else (0, 18)):
        A = esp.ExtendableSparseMatrix(N, N)
        A.debug_force_path(force)
        A.generate_fdrand(n, n, n, seed=0x5EED0002, rand_mode=1)
        A.flush()
        assert A.debug_last_partition() == (1 if force == 16 else 4)
        assert A.debug_last_key_bytes() == 4
        if n == 256:   # segments of 256 columns x 12 updates: the small variant of the bucket kernel (18: the regular one)
            assert A.debug_last_local_small() == (1 if force == 0 else 0)
        arrs = hip_arrays(A)
        assert len(arrs[1]) == int(d["nnz"])
        assert gu.digest(*arrs) == d["csc"], (n, force)
        del A, arrs


def test_more_than_32_key_bits_below_the_prefix(esp):
    """A 256^3 stencil leaves exactly 32 key bits below its 16-bit prefix (4-byte keys everywhere); a larger problem -- 322^3: 33
    bits -- takes the FINE partition (round 6): one more prefix bit, so that the rest fits 4-byte keys, while the bucket kernel
    still takes the planned segments (two buckets each, an entry's bucket told by its position).  Device results that must agree
    bit for bit with each other and with the oracle's digest at this size -- the plain producer + the flush's own partition (packed keys), the producer-side
    partition with the fine plan (4-byte keys) and with the hook that forbids it (41: packed keys), the same stream as resident
    triplets through esp_append_device (fine plan, then its repetition over the kept run lists), and the shard producer through the
    group API (its own range as 4-byte keys of a fine partition; with hook 41 packed keys)."""
    import ctypes as C
    torch = pytest.importorskip("torch")
    n = 322
    N = n ** 3
    E = 12 * n * n * (n - 1) + 6 * n * n
    seen = {}
    for name, force, kb, part in (("flush_partition", 16, 8, 1), ("producer_fine", 0, 4, 4), ("producer_packed", 41, 8, 4)):
        A = esp.ExtendableSparseMatrix(N, N, capacity_hint=E)
        A.debug_force_path(force)
        for it in range(2 if force == 0 else 1):        # (the second assembly reuses the generator's plan, fine bits included)
            A.reset()
            A.generate_fdrand(n, n, n, seed=0x5EED0002, rand_mode=1)
            A.flush()
            assert A.debug_last_key_bytes() == kb, (name, A.debug_last_key_bytes())
            assert A.debug_last_partition() == part
        seen[name] = gu.digest(*hip_arrays(A))
        del A
    # the stream as a caller's resident triplets (stream order: the plain producer's packed keys, unpacked)
    keys = torch.empty(E, dtype=torch.int64, device="cuda")
    vals = torch.empty(E, dtype=torch.float64, device="cuda")
    G = esp.ExtendableSparseMatrix(N, N, capacity_hint=E)
    G.debug_force_path(16)
    G.generate_fdrand(n, n, n, seed=0x5EED0002, rand_mode=1)
    offs = (C.c_int64 * 2)()
    rbits, cbits = C.c_int32(), C.c_int32()
    G._d.ck(G._d.lib.esp_key_layout(G._d.h, C.byref(rbits), C.byref(cbits)))
    G._d.ck(G._d.lib.esp_shard_export(G._d.h, 1, C.c_void_p(keys.data_ptr()), C.c_void_p(vals.data_ptr()), offs))
    rb = rbits.value
    rows = ((keys >> 2) & ((1 << rb) - 1)) + 1
    cols = (keys >> (2 + rb)) + 1
    del keys, G
    T = esp.ExtendableSparseMatrix(N, N, capacity_hint=E)
    for it in range(2):
        T.reset()
        T.append_device(esp.ESP_UPDATE, rows, cols, vals)
        T.flush()
        assert T.debug_last_key_bytes() == 4 and T.debug_last_partition() == 4, (it, T.debug_last_key_bytes(), T.debug_last_partition())
        assert T.debug_last_plan_reused() == it
        seen["triplets_%d" % it] = gu.digest(*hip_arrays(T))
    del T, rows, cols, vals
    # the shard producer: its tables 2^fb times finer than the exchange's digits, the own range 4-byte keys (third assembly: over
    # the kept plan and the kept owner ranges); with hook 41: packed keys
    for name, force, kb in (("shard_producer", 0, 4), ("shard_producer_packed", 41, 8)):
        SA = esp.GroupShardedMatrix(N, N, nranks=1, rank=0, capacity_hint=E)
        A = SA.local
        A.debug_force_path(force)
        for it in range(3 if force == 0 else 2):   # (the second assembly finds the plan of the first flush: the producer partitions)
            A.reset()
            A.generate_fdrand(n, n, n, seed=0x5EED0002, rand_mode=1)
            SA.flush()
            if it > 0:
                assert A.debug_last_shard_source() == 2 and A.debug_last_key_bytes() == kb, (name, it, A.debug_last_key_bytes())
        seen[name] = gu.digest(*hip_arrays(A))
        del SA, A
    assert len(set(seen.values())) == 1, seen
    # ... and they are the oracle's bits (tests/golden/make_digests_large.py fd_322: the CPU restatement at this size, made once)
    assert seen["producer_fine"] == gu.digests("digests_large.txt")["fd_322_m1"]["csc"]


def test_mt_per_entry_digest(esp):
    """The per-entry form of the reference's multi-threaded assembly at bench size (test/femtools.jl:88-107: 16 tasks x 2 10^6
    updateindex! / rawupdateindex! calls with their tid, ONE flush! = Base.sum(xmatrices, csc)): esp_flush_sum's general path --
    every buffer's own fold, side by side on the library's host pool -- against the oracle's MT wrapper (tests/golden: mtgen_4M_p16)."""
    import ctypes as C
    d = gu.digests("digests_large.txt")["mtgen_4M_p16"]
    n, p = 4000000, 16
    xs = [esp.SparseMatrixHIPCOO(n, n) for _ in range(p)]
    home = esp.SparseMatrixHIPCOO(n, n)
    streams = gu.mt_per_entry_streams(n, p)
    for rnd in range(2):                                 # (fresh handles, then warm ones)
        home._d.ck(home._d.lib.esp_reset(home._d.h))
        for t, (I, J, V, K) in enumerate(streams):
            xs[t].append(0, I, J, V, kinds=K)
        arr = (C.c_void_p * p)(*[x._d.h for x in xs])
        z, ch = C.c_int64(), C.c_int32()
        home._d.ck(home._d.lib.esp_flush_sum(home._d.h, arr, p, C.byref(z), C.byref(ch)))
        assert z.value == int(d["nnz"])
        assert gu.digest(*home._d.get_csc().arrays()) == d["csc"], rnd


def test_config3_digest_128(esp):
    """BASELINE config 3 at 128^3: stored stencil CSC + new second-neighbour positions + the full stream again, one
    flush through the routed fold and the merge-path join; digest of the oracle's result."""
    n = 128
    N = n ** 3
    d = gu.digests("digests_large.txt")["cfg3_%d" % n]
    I2, J2, V2 = gu.cfg3_new_positions(n)
    for order in ("append_first", "generate_first", "generate_first_19", "generate_first_29", "generate_first_40"):
        A = esp.ExtendableSparseMatrix(N, N)
        A.generate_fdrand(n, n, n, seed=0x5EED0002, rand_mode=1)
        A.flush()
        if order.endswith("_40"):
            A.debug_force_path(40)                   # (never the rebuild: the bucket kernel against the stored columns + the join)
        if order.endswith("_19"):
            A.debug_force_path(19)                   # (no batch + tail flush: packed keys, the ordinary partition)
        if order.endswith("_29"):
            A.debug_force_path(29)                   # (the tail is copied to the front, not partitioned where it lies)
        if order == "append_first":
            A.append(UPDATE, I2, J2, V2)
            A.generate_fdrand(n, n, n, seed=0x5EED0012, rand_mode=1)
        else:
            # (different call order, same result: the new positions and the stencil's never coincide; the bench's call:
            # triplets resident on the device)
            import torch
            A.generate_fdrand(n, n, n, seed=0x5EED0012, rand_mode=1)
            A.append_device(UPDATE, torch.from_numpy(I2).cuda(), torch.from_numpy(J2).cuda(), torch.from_numpy(V2).cuda())
        A.flush()
        # (the bench's order: the producer's batch is flushed as it is, the new couplings as a flush of their own)
        # (8: the new couplings were partitioned as they were appended; 29 reads them as a packed tail)
        assert A.debug_last_partition() == {"generate_first": 8, "generate_first_29": 6}.get(order, A.debug_last_partition()), order
        assert (A.debug_last_partition() in (6, 8)) == (order in ("generate_first", "generate_first_29", "generate_first_40")), (order, A.debug_last_partition())
        # (the couplings behind the batch REBUILD the matrix -- the stored entries as the first piece of a fresh flush: flush_rebuild)
        assert A.debug_last_rebuild() == (1 if order == "generate_first" else 0), (order, A.debug_last_rebuild())
        arrs = hip_arrays(A)
        assert len(arrs[1]) == int(d["nnz"])
        assert gu.digest(*arrs) == d["csc"], order


def test_config3_digest_bench_size(esp):
    """BASELINE config 3 at the size bench.py times (256^3 stored + the stream again + 33 M new positions): the call
    order of bench.py's extra.configs (batch flushed by itself, then the tail with the column-tiled join)."""
    n = 256
    N = n ** 3
    d = gu.digests("digests_large.txt")["cfg3_%d" % n]
    I2, J2, V2 = gu.cfg3_new_positions(n)
    A = esp.ExtendableSparseMatrix(N, N)
    A.generate_fdrand(n, n, n, seed=0x5EED0002, rand_mode=1)
    A.flush()
    A.generate_fdrand(n, n, n, seed=0x5EED0012, rand_mode=1)
    A.append(UPDATE, I2, J2, V2)
    A.flush()
    assert A.debug_last_partition() in (6, 8)   # (8: the host append arrived as one chunk and was partitioned as it came)
    assert A.debug_last_rebuild() == (1 if A.debug_last_partition() == 8 else 0)
    arrs = hip_arrays(A)
    assert len(arrs[1]) == int(d["nnz"])
    assert gu.digest(*arrs) == d["csc"]
    # ... and a third assembly over the rebuilt matrix: every update hits now (the in-place kernels again)
    A.generate_fdrand(n, n, n, seed=0x5EED0012, rand_mode=1)
    A.append(UPDATE, I2, J2, V2)
    A.flush()
    assert A.debug_last_rebuild() == 0 or A.nnz() == int(d["nnz"])
    assert A.nnz() == int(d["nnz"])


@pytest.mark.parametrize("dim,npd", [(2, 3163), (3, 216)])
def test_fem_digest_bench_size(esp, dim, npd):
    """BASELINE config 4 at the sizes bench.py times (10^7 DoF, random cell order; the oracle was fed the stream in
    chunks, tests/golden/make_digests_large.py): item partition + expansion + group tier of the bucket kernel."""
    d = gu.digests("digests_large.txt")["fem%dd_%d_o1" % (dim, npd)]
    nn = npd ** dim
    A = esp.ExtendableSparseMatrix(nn, nn)
    A.generate_fem(dim, npd, seed=0x5EED0004, order_mode=1)
    A.flush()
    assert A.debug_last_partition() == 4 and A.debug_last_lazy_items() == 1
    arrs = hip_arrays(A)
    assert len(arrs[1]) == int(d["nnz"])
    assert gu.digest(*arrs) == d["csc"]


@pytest.mark.parametrize("dim,npd", [(2, 1000), (3, 64), (2, 3163), (3, 216)])
@pytest.mark.parametrize("node_mode", [0, 1])
def test_elements_digest(esp, dim, npd, node_mode):
    """esp_append_elements at 10^6 .. 10^7 DoF (the last two: the sizes bench.py times), random cell order, element arrays
    made on the device (esp_generate_fem_mesh).  Natural node numbering: the stream is generate_fem's, the pin the same;
    permuted numbering (a mesh that owes nothing to grid arithmetic): pins made by the oracle from its own arrays through
    the loops of femtools.jl:61-69 (tests/golden/make_digests_large.py: elements_chunked)."""
    import torch
    tag = ("fem%dd_%d_o1" if node_mode == 0 else "elem%dd_%d_p1") % (dim, npd)
    d = gu.digests("digests_large.txt")[tag]
    nn, nloc = npd ** dim, dim + 1
    q = npd - 1
    nc = 2 * q * q if dim == 2 else 6 * q ** 3
    A = esp.ExtendableSparseMatrix(nn, nn)
    cn = torch.empty((nc, nloc), dtype=torch.int64, device="cuda")
    em = torch.empty((nc, nloc, nloc), dtype=torch.float64, device="cuda")
    dg = torch.empty((nc, nloc), dtype=torch.float64, device="cuda")
    A.generate_fem_mesh(dim, npd, cn, em, dg, seed=0x5EED0004, order_mode=1, node_mode=node_mode, node_seed=0x5EED0014)
    A.append_elements(cn, em, dg)
    A.flush()
    assert A.debug_last_partition() == 4
    # natural numbering: the group-tier kernel with three workgroups per CU (2); permuted numbering of more than 2^18 nodes:
    # its wide form (3: full rows in LDS, every column run sorted twice) -- not local_k's radix tier
    assert A.debug_last_local_small() == (3 if (node_mode == 1 and nn > (1 << 18)) else 2), A.debug_last_local_small()
    # (the fused bucket kernel formed the updates from the item records -- unless it refused the segments for their rows:
    # then the items were expanded and the wide form took the entries)
    assert A.debug_last_lazy_items() == (0 if (node_mode == 1 and nn > (1 << 18)) else 1)
    arrs = hip_arrays(A)
    assert len(arrs[1]) == int(d["nnz"])
    assert gu.digest(*arrs) == d["csc"]
    if node_mode == 1 and npd in (1000, 64):
        # the same without the wide form (force_path 33: local_k's kernels), and a second assembly on the handle that has
        # learnt (straight to the wide form)
        B = esp.ExtendableSparseMatrix(nn, nn)
        B.debug_force_path(33)
        B.append_elements(cn, em, dg)
        B.flush()
        assert B.debug_last_local_small() != 3
        assert gu.digest(*hip_arrays(B)) == d["csc"]
        A.reset()
        A.append_elements(cn, em, dg)
        A.flush()
        assert gu.digest(*hip_arrays(A)) == d["csc"]


@pytest.mark.parametrize("dim,npd", [(2, 1000), (3, 64)])
@pytest.mark.parametrize("order", [0, 1])
def test_fem_digest(esp, dim, npd, order):
    """BASELINE config 4 at 10^6 / 2.6 10^5 DoF, natural (producer-side partition by run lists) and random cell order
    (item partition + expansion, femitems.hpp)."""
    d = gu.digests("digests_large.txt")["fem%dd_%d_o%d" % (dim, npd, order)]
    nn = npd ** dim
    A = esp.ExtendableSparseMatrix(nn, nn)
    A.generate_fem(dim, npd, seed=0x5EED0004, order_mode=order)
    A.flush()
    assert A.debug_last_partition() == 4   # natural order: run lists; random: item partition (femitems.hpp)
    arrs = hip_arrays(A)
    assert len(arrs[1]) == int(d["nnz"])
    assert gu.digest(*arrs) == d["csc"]
    # random order: the batch stayed a list of sorted items and the bucket kernel formed the updates itself (group3_items.hpp);
    # force_path 39: the expansion at append time, as before -- the same CSC
    # (single-word item records only: the cell's number must fit below the column bits of a key -- 3-D, 64^3 nodes: it does not)
    q = npd - 1
    ncells = 2 * q * q if dim == 2 else 6 * q ** 3
    single = ncells <= 1 << (max(1, (nn - 1).bit_length()) + 2)
    assert A.debug_last_lazy_items() == (1 if (order == 1 and single) else 0)
    if order == 1:
        B = esp.ExtendableSparseMatrix(nn, nn)
        B.debug_force_path(39)
        B.generate_fem(dim, npd, seed=0x5EED0004, order_mode=order)
        B.flush()
        assert B.debug_last_partition() == 4 and B.debug_last_lazy_items() == 0
        assert gu.digest(*hip_arrays(B)) == d["csc"]


@pytest.mark.parametrize("dims", [(4096, 300, 1), (300, 4096, 1), (2000001, 1, 1), (37, 41, 1013), (1021, 3, 509), (2, 2, 400000),
                                  (129, 127, 131)])
def test_producer_partition_odd_grids(esp, orc, dims):
    """Producer-side partition on grids whose lines, planes and buckets do not line up with the 256-node chunks
    (1-D and 2-D stencils, prime extents, two-node lines): the oracle's bits, and the same bits without it."""
    nx, ny, nz = dims
    N = nx * ny * nz
    O = orc.fdrand(nx, ny, nz, rand_mode=1, seed=99, style=orc.KIND_UPDATE)
    want = O.arrays()
    seen = set()
    for force in (0, 16):
        A = esp.ExtendableSparseMatrix(N, N)
        A.debug_force_path(force)
        A.generate_fdrand(nx, ny, nz, seed=99, rand_mode=1)
        A.flush()
        seen.add(A.debug_last_partition())
        assert_csc_equal(hip_arrays(A), want, "%s force %d" % (dims, force))
    assert 4 in seen, seen


@pytest.mark.parametrize("focus,seed,seconds", [("", 11, 20), ("k32", 12, 12), ("elements", 13, 15)])
def test_bounded_fuzz(focus, seed, seconds):
    """tests/fuzz_parity.py (random shapes, kinds, orders, flush sequences, forced paths; every result against the
    oracle bit for bit) with fixed seeds and a bounded budget."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    if focus:
        env["ESP_FUZZ_FOCUS"] = focus
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_parity.py"), str(seconds), str(seed)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "fuzz ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


# ------------------------------------------------------------------ the C group API (esp_group_*)
def _group_ranks_run(esp, orc, world, deal, n=44, rounds=3):
    """W ranks as threads on one GPU, each with its own handle and esp_group (transport: the callback table of
    tests/threaddist.py::ThreadComm): the exchange and its policy run inside the library.  The stitched CSC must equal
    ONE oracle buffer fed the ranks' streams in rank order."""
    from threaddist import ThreadComm, run_ranks
    nx = ny = n
    nzg = n * world
    N = nx * ny * nzg
    nodes = nx * ny * n
    seeds = [77, 78, 79]
    streams = [orc.fdrand_stream(nx, ny, nzg, rand_mode=1, seed=seeds[r]) for r in range(rounds)]
    E = len(streams[0][0])
    kinds = np.where(np.arange(E) % 5 == 0, RAW, UPDATE).astype(np.uint8)
    if deal == "slab":
        l = np.arange(N)
        i, j, k = l % nx + 1, (l // nx) % ny + 1, l // (nx * ny) + 1
        per = (4 * (i < nx) + ((i == 1) | (i == nx)) + 4 * (j < ny) + ((j == 1) | (j == ny))
               + 4 * (k < nzg) + ((k == 1) | (k == nzg)))
        off = np.concatenate([[0], np.cumsum(per)])
        sel = [slice(int(off[r * nodes]), int(off[(r + 1) * nodes])) for r in range(world)]
        kinds[:] = UPDATE
    else:
        chunk = np.arange(E) // 4096
        sel = [(chunk % world) == r for r in range(world)]
    perm1 = np.random.default_rng(9).permutation(int(np.count_nonzero(sel[1])) if deal != "slab" else 1)

    def rank_stream(rank, rnd):
        I, J, V = streams[rnd]
        Ii, Jj, Vv, kk = I[sel[rank]], J[sel[rank]], V[sel[rank]], kinds[sel[rank]]
        if deal == "shuffled_rank1" and rank == 1:
            Ii, Jj, Vv, kk = Ii[perm1], Jj[perm1], Vv[perm1], kk[perm1]
        return Ii, Jj, Vv, kk

    comm = ThreadComm(world, esp._lib)

    def body(rank, dist):
        holder = {}
        table = comm.table(rank, lambda: holder["A"].local._d.h)
        A = esp.GroupShardedMatrix(N, N, nranks=world, rank=rank, comm=table)
        holder["A"] = A
        hist = []
        for rnd in range(rounds):
            if deal == "slab":
                A.local.generate_fdrand_range(nx, ny, nzg, rank * nodes, (rank + 1) * nodes, seed=seeds[rnd], rand_mode=1)
            else:
                Ii, Jj, Vv, kk = rank_stream(rank, rnd)
                A.local.append(0, Ii, Jj, Vv, kinds=kk)
            A.flush()
            hist.append((A.last_exchange, A.local.debug_last_partition()))
            if deal == "slab":
                # from the second assembly on the generator partitions by (owner, digit) itself (esp_shard_plan, set by
                # the previous esp_group_flush): esp_shard_partition has nothing left to move
                assert A.local.debug_last_shard_source() == (1 if rnd == 0 else 2), (rnd, A.local.debug_last_shard_source())
                # ... its own range with 4-byte keys; the neighbours' packed keys are narrowed as the bucket kernel loads them
                # (every entry is an UPDATE: the 4-byte-key kernel with the UPDATE-only fold, local.hpp KEYS 5)
                if rnd > 0:
                    assert A.local.debug_last_key_bytes() == 4 and A.local.debug_last_fold_update(), (rnd, A.local.debug_last_key_bytes())
        total = A.nnz()
        piece = A.local_slice()
        lo, hi = A.column_range()
        assert (lo - 1, hi) == esp.owner_ranges(N, world)[rank]
        return hist, piece, total

    outs = run_ranks(world, body)
    if comm.errors:
        raise comm.errors[0]
    O = orc.ExtendableSparseMatrix(N, N)
    for rnd in range(rounds):
        for rank in range(world):
            Ii, Jj, Vv, kk = rank_stream(rank, rnd)
            O.apply(kk, Ii, Jj, Vv)
        O.flush()
    G = esp.GroupShardedMatrix.stitch(N, N, [o[1] for o in outs], outs[0][2])
    assert_csc_equal(G.arrays(), O.arrays())
    assert all(o[2] == O.nnz() for o in outs)
    return [o[0] for o in outs]


@pytest.mark.parametrize("world,deal", [(2, "slab"), (3, "slab"), (2, "scrambled"), (3, "scrambled")])
def test_group_api_partitioned_exchange(esp, orc, world, deal):
    hist = _group_ranks_run(esp, orc, world, deal)
    for h in hist:
        assert h == [("partitioned", 7)] * 3, h


@pytest.mark.parametrize("force,kind", [(0, UPDATE), (0, RAW), (41, UPDATE)])
def test_group_api_fine_partition_two_ranks(esp, force, kind):
    """Two ranks (threads, one GPU) whose shards leave more than 32 key bits below the plan's prefix (322 x 322 x 162 nodes): from the
    second assembly on the producers cut every digit of the plan in two or four buckets, their own ranges hold 4-byte keys, the neighbour's packed
    keys are narrowed by the bucket kernel (boundary segments: two pieces) -- what an 8-GPU run of 512^3 does with 35 bits.  No
    CPU oracle at this size: the stitched CSC must be the unsharded handle's, bit for bit; hook 41 (no fine partition): packed keys;
    rawupdateindex! calls (kind RAW): the bucket kernel's generic fold over pieces of which one holds 4-byte keys (KEYS 4)."""
    from threaddist import ThreadComm, run_ranks
    world, nx, ny, n = 2, 322, 322, 81      # (a shard just above 2^23 columns, 25 row bits: 33 or 34 bits below the plan's prefix)
    nzg = n * world
    N = nx * ny * nzg
    nodes = nx * ny * n
    R = esp.ExtendableSparseMatrix(N, N)
    for seed in (77, 78, 79):            # (three assemblies onto the same matrix, like the ranks below)
        R.generate_fdrand(nx, ny, nzg, seed=seed, rand_mode=1, kind=kind)
        R.flush()
    want = gu.digest(*hip_arrays(R))
    want_nnz = R.nnz()
    del R
    comm = ThreadComm(world, esp._lib)

    def body(rank, dist):
        holder = {}
        table = comm.table(rank, lambda: holder["A"].local._d.h)
        A = esp.GroupShardedMatrix(N, N, nranks=world, rank=rank, comm=table)
        holder["A"] = A
        A.local.debug_force_path(force)
        for rnd, seed in enumerate((77, 78, 79)):
            A.local.generate_fdrand_range(nx, ny, nzg, rank * nodes, (rank + 1) * nodes, seed=seed, rand_mode=1, kind=kind)
            A.flush()
            assert A.last_exchange == "partitioned" and A.local.debug_last_partition() == 7
            assert bool(A.local.debug_last_fold_update()) == (kind == UPDATE)
            if rnd > 0:
                assert A.local.debug_last_shard_source() == 2
                assert A.local.debug_last_key_bytes() == (4 if force == 0 else 8), (rnd, A.local.debug_last_key_bytes())
        return A.local_slice(), A.nnz()

    outs = run_ranks(world, body)
    if comm.errors:
        raise comm.errors[0]
    assert outs[0][1] == want_nnz
    G = esp.GroupShardedMatrix.stitch(N, N, [o[0] for o in outs], outs[0][1])
    assert gu.digest(*G.arrays()) == want


def test_group_api_falls_back_by_consensus(esp, orc):
    hist = _group_ranks_run(esp, orc, 2, "shuffled_rank1")
    for h in hist:
        assert [x[0] for x in h] == ["inplace"] * 3, h


def test_group_api_world1_rccl(esp, orc):
    """The RCCL transport of the library itself (ncclCommInitRank from a unique id, world size 1): fresh build, then
    re-assembly with new entries; global nnz / column range / stitched CSC."""
    n = 40
    N = n ** 3
    uid = esp.GroupShardedMatrix.unique_id()
    assert len(uid) == 128
    A = esp.GroupShardedMatrix(N, N, nranks=1, rank=0, unique_id=uid)
    O = orc.ExtendableSparseMatrix(N, N)
    rng = np.random.default_rng(2)
    for rnd in range(2):
        A.local.generate_fdrand(n, n, n, seed=5 + rnd, rand_mode=1)
        I, J, V = orc.fdrand_stream(n, n, n, rand_mode=1, seed=5 + rnd)
        O.apply(np.full(len(I), UPDATE, np.uint8), I, J, V)
        Ix, Jx, Vx = rng.integers(1, N + 1, 3000), rng.integers(1, N + 1, 3000), rng.standard_normal(3000)
        A.local.append(RAW, Ix, Jx, Vx)
        O.apply(np.full(3000, RAW, np.uint8), Ix, Jx, Vx)
        A.flush()
        O.flush()
        assert A.nnz() == O.nnz() and A.column_range() == (1, N)
        c0, c1, cp, rv, nz = A.local_slice()
        assert_csc_equal((cp, rv, nz), O.arrays())


def test_group_rccl_loopback(esp, orc):
    """The library's RCCL transport really moving data on a one-GPU box (esp_debug_group_loopback): a single-rank group
    sends its own partitioned ranges to itself through rccl_alltoallv (grouped ncclSend / ncclRecv to self on the handle's
    stream), wipes them and restores them from what arrived; the flush's agreements run as ncclAllGather on the second
    stream.  Partitioned exchange (a flush's own pass, then the producer's), in-place exchange (a shuffled stream:
    consensus fall-back), re-assembly with new entries -- against the oracle."""
    n = 64
    N = n ** 3
    A = esp.GroupShardedMatrix(N, N, nranks=1, rank=0)
    A.debug_loopback(True)
    O = orc.ExtendableSparseMatrix(N, N)
    rng = np.random.default_rng(3)
    kinds_seen = []
    for rnd in range(5):
        moved = 0
        if rnd < 4:                                # the stencil's pre-sorted stream: partitioned exchange
            A.local.generate_fdrand(n, n, n, seed=5 + rnd, rand_mode=1)
            I, J, V = orc.fdrand_stream(n, n, n, rand_mode=1, seed=5 + rnd)
            O.apply(np.full(len(I), UPDATE, np.uint8), I, J, V)
            moved += len(I)
        if rnd >= 2:                               # ... with a shuffled tail / a shuffled stream by itself: in place
            cnt = 3000 if rnd < 4 else 200000
            Ix, Jx, Vx = rng.integers(1, N + 1, cnt), rng.integers(1, N + 1, cnt), rng.standard_normal(cnt)
            A.local.append(RAW, Ix, Jx, Vx)
            O.apply(np.full(cnt, RAW, np.uint8), Ix, Jx, Vx)
            moved += cnt
        A.flush()
        O.flush()
        kinds_seen.append(A.last_exchange)
        assert A.loopback_bytes() >= 12 * moved, (rnd, A.loopback_bytes(), moved)
        c0, c1, cp, rv, nz = A.local_slice()
        assert_csc_equal((cp, rv, nz), O.arrays(), "round %d" % rnd)
    assert kinds_seen[:2] == ["partitioned", "partitioned"] and "inplace" in kinds_seen, kinds_seen


def test_group_rccl_loopback_large_message(esp):
    """... and with messages above the 1 GiB of one round (256^3: 1.6 GB of keys, 1.6 GB of values per flush): the
    device CSC's digest equals the oracle's pin of the bench configuration."""
    n = 256
    N = n ** 3
    d = gu.digests("digests_large.txt")["fd_%d_m1" % n]
    A = esp.GroupShardedMatrix(N, N, nranks=1, rank=0, capacity_hint=12 * n * n * (n - 1) + 6 * n * n)
    A.debug_loopback(True)
    for rnd in range(2):                     # (the second assembly: the producer partitions for the exchange itself)
        A.local.reset()
        A.local.generate_fdrand(n, n, n, seed=0x5EED0002, rand_mode=1)
        A.flush()
        assert A.loopback_bytes() == 16 * (12 * n * n * (n - 1) + 6 * n * n), A.loopback_bytes()   # two messages of 1.6 GB: two rounds each
        arrs = hip_arrays(A.local)
        assert len(arrs[1]) == int(d["nnz"])
        assert gu.digest(*arrs) == d["csc"], rnd


def test_jacobi_and_ilu0_setup(esp, orc):
    """set-up of the point preconditioners on the device CSC (jacobi.jl:5-12, ilu0.jl:8-41) == the
    oracle's literal restatement, bit for bit; missing diagonal: Inf (Jacobi) / an error (ILU0: undefined idiag)."""
    n = 40
    N = n ** 3
    A = esp.fdrand(n, n, n, rand_mode=1, seed=11)
    O = orc.fdrand(n, n, n, rand_mode=1, seed=11, style=orc.KIND_UPDATE)
    C0 = O.sparse()
    assert np.array_equal(bits(A.jacobi()), bits(C0.jacobi()))
    xd, idg = A.ilu0()
    xo, io = C0.ilu0()
    assert np.array_equal(bits(xd), bits(xo)) and np.array_equal(idg, io)
    # device-resident outputs
    import ctypes
    import torch
    tx = torch.empty(N, dtype=torch.float64, device="cuda")
    ti = torch.empty(N, dtype=torch.int64, device="cuda")
    d = A._d
    d.ck(d.lib.esp_ilu0_setup(d.h, ctypes.c_void_p(tx.data_ptr()), ctypes.c_void_p(ti.data_ptr()), 1))
    assert np.array_equal(bits(tx.cpu().numpy()), bits(xo)) and np.array_equal(ti.cpu().numpy(), io)
    # rectangular -> error; a column without a diagonal entry
    B = esp.ExtendableSparseMatrix(5, 5)
    B.append(UPDATE, [1, 2, 4, 5, 1], [1, 2, 4, 5, 3], [2.0, 4.0, 8.0, 16.0, 1.0])
    inv = B.jacobi()
    assert list(inv[[0, 1, 3, 4]]) == [0.5, 0.25, 0.125, 0.0625] and np.isinf(inv[2])
    with pytest.raises(esp.EspError):
        B.ilu0()
    with pytest.raises(esp.EspError):
        esp.ExtendableSparseMatrix(4, 5).jacobi()


@pytest.mark.parametrize("force", [0, 17])
def test_join_dense_columns_empty_columns_and_rectangular(esp, orc, force):
    """The join with an existing CSC on shapes that stress its column tiles: columns with thousands of stored and new
    entries next to long runs of empty columns, new entries before / between / behind the stored rows of a column, a
    wide rectangular matrix, PLUS mode (csc + buffer)."""
    rng = np.random.default_rng(17)
    for (m, n) in ((5000, 1000), (300, 70000), (2 ** 20, 513)):
        A = esp.ExtendableSparseMatrix(m, n)
        A.debug_force_path(force)
        O = orc.ExtendableSparseMatrix(m, n)
        for rnd in range(3):
            cnt = 60000
            heavy = rng.choice(n, 3, replace=False) + 1
            J = np.where(rng.random(cnt) < 0.5, rng.choice(heavy, cnt), rng.integers(1, n + 1, cnt))
            J[rng.random(cnt) < 0.3] = rng.integers(max(1, n // 2), min(n, n // 2 + 40) + 1, int((rng.random(cnt) < 0.3).sum()) or 1)[0]
            I = rng.integers(1, m + 1, cnt)
            V = rng.standard_normal(cnt)
            kinds = rng.integers(0, 3, cnt).astype(np.uint8)
            A.append(0, I, J, V, kinds=kinds)
            O.apply(kinds, I, J, V)
            A.flush()
            O.flush()
            assert_csc_equal(hip_arrays(A), O.arrays(), "%dx%d round %d" % (m, n, rnd))
    # PLUS mode through the buffer type: csc + x
    m, n = 700, 900
    C0 = orc.ExtendableSparseMatrix(m, n)
    I, J, V = rng.integers(1, m + 1, 9000), rng.integers(1, n + 1, 9000), rng.standard_normal(9000)
    C0.apply(np.full(9000, RAW, np.uint8), I, J, V)
    cp, rv, nz = C0.arrays()
    X = esp.SparseMatrixHIPCOO(m, n)
    X._d.lib.esp_debug_force_path(X._d.h, force)
    L = orc.SparseMatrixLNK(m, n)
    I2, J2, V2 = rng.integers(1, m + 1, 9000), rng.integers(1, n + 1, 9000), rng.standard_normal(9000)
    X.append(UPDATE, I2, J2, V2)
    for i, j, v in zip(I2, J2, V2):
        L.updateindex(orc.OP_ADD, float(v), int(i), int(j))
    got = X + esp.SparseMatrixCSC(m, n, cp, rv, nz)
    want = L + orc.CSC(m, n, cp, rv, nz)
    assert_csc_equal(got.arrays(), want.arrays())


def test_small_bucket_kernel_slow_tier_on_long_runs(esp, orc):
    """The small variant of the bucket kernel (three workgroups per CU, no radix tier) is chosen from what the host
    knows BEFORE the flush (segments of at most 3072 entries over at most 256 columns, few entries per column on average
    or what the handle's last flush saw).  Column runs longer than its register tiers take go through its slow tier
    (insertion sort of the run in LDS + sequential fold) and send the handle's next flushes to the regular kernel: the
    oracle's bits both times, on a fresh matrix and over a stored pattern."""
    rng = np.random.default_rng(31)
    m, n = 2 ** 20, 2 ** 16
    A = esp.ExtendableSparseMatrix(m, n)
    O = orc.ExtendableSparseMatrix(m, n)
    per_col, ncols_used = 50, 19000
    for rnd in range(3):
        cols = np.sort(rng.choice(n, ncols_used, replace=False)) + 1
        J = np.repeat(cols, per_col)
        # a few rows per column, many duplicates: long column runs, few records
        I = np.clip(J[:, None] * 7 % m + rng.integers(0, 5, (len(J), 1)), 1, m).ravel()
        V = rng.standard_normal(len(J))
        kinds = rng.integers(0, 3, len(J)).astype(np.uint8) if rnd == 2 else np.full(len(J), UPDATE, np.uint8)
        if rnd == 2:
            A.append(0, I, J, V, kinds=kinds)
        else:
            A.append(UPDATE, I, J, V)
        O.apply(kinds, I, J, V)
        A.flush()
        O.flush()
        assert A.debug_last_path() == 1
        # first flush: no history, 14 entries per column on average -> the small variant, whose slow tier serves the
        # runs of 50; it reports them, so the later flushes (over the stored pattern) take the regular kernel
        assert A.debug_last_local_small() == (1 if rnd == 0 else 0), (rnd, A.debug_last_local_small())
        assert_csc_equal(hip_arrays(A), O.arrays(), "round %d" % rnd)
    # the same through the routed fold of the small variant: a stored pattern, short runs, then one flush with long runs
    B = esp.ExtendableSparseMatrix(m, n)
    Ob = orc.ExtendableSparseMatrix(m, n)
    used = []
    for rnd, pc in enumerate((8, 8, 50)):
        cols = np.sort(rng.choice(n, 60000 if pc == 8 else 19000, replace=False)) + 1
        J = np.repeat(cols, pc)
        I = np.clip(J[:, None] * 7 % m + rng.integers(0, 5, (len(J), 1)), 1, m).ravel()
        V = rng.standard_normal(len(J))
        B.append(UPDATE, I, J, V)
        Ob.apply(np.full(len(J), UPDATE, np.uint8), I, J, V)
        B.flush()
        Ob.flush()
        used.append(B.debug_last_local_small())
        assert_csc_equal(hip_arrays(B), Ob.arrays(), "routed round %d" % rnd)
    # (the partition plan of a later flush may cut wider segments than the small variant takes; the first flush and the
    # one with the long runs -- history says short runs -- use it)
    assert used[0] == 1 and used[2] == 1, used


def test_no_append_between_shard_assemble_and_flush(esp, orc):
    """Between esp_shard_assemble and esp_flush the pending entries are the caller's receive buffers (the handle's count
    is their logical total): an append must be refused (ESP_ERR_STATE), the flush afterwards gives the oracle's bits."""
    import ctypes as C
    n = 48
    N = n ** 3
    A = esp.ExtendableSparseMatrix(N, N)
    d = A._d
    A.generate_fdrand(n, n, n, seed=9, rand_mode=1)
    E = A.nnznew()
    ok = C.c_int32()
    pk, pv, pc = C.c_void_p(), C.c_void_p(), C.c_void_p()
    eoff = np.zeros(2, np.int64)
    nb = C.c_int64()
    d.ck(d.lib.esp_set_column_window(d.h, 1, N))
    d.ck(d.lib.esp_shard_partition(d.h, 1, 0, E, C.byref(ok), C.byref(pk), C.byref(pv), C.byref(pc),
                                   eoff.ctypes.data_as(C.c_void_p), C.byref(nb)))
    assert ok.value == 1 and eoff[1] == E
    none = (C.c_void_p * 1)()
    ne = np.zeros(1, np.int64)
    d.ck(d.lib.esp_shard_assemble(d.h, none, none, none, ne.ctypes.data_as(C.c_void_p), C.byref(ok)))
    assert ok.value == 1
    with pytest.raises(esp.EspError):
        A.append(UPDATE, [1], [1], [1.0])
    A.flush()
    assert A.debug_last_partition() == 7
    O = orc.fdrand(n, n, n, rand_mode=1, seed=9, style=orc.KIND_UPDATE)
    assert_csc_equal(hip_arrays(A), O.arrays())


@pytest.mark.parametrize("self_rank", [0, 63])
def test_shard_assemble_with_64_shards(esp, orc, self_rank):
    """The most shards the partitioned exchange takes (64): one handle plays shard `self_rank`, nothing arrives from the
    others (their ranges are sent and dropped here), the local flush must give the oracle's columns of that shard bit for bit.
    (Round 6: the table kernel's summary -- entries per source, start of the own range -- has P + 1 words; with 64 shards the
    last one used to be the same word as the first piece start.)"""
    import ctypes as C
    import torch
    n = 128          # (64 columns per digit: a chunk of the stencil's stream stays within the run limit of the run-based partition)
    N = n ** 3
    P = 64
    A = esp.ExtendableSparseMatrix(N, N)
    d = A._d
    I, J, V = orc.fdrand_stream(n, n, n, rand_mode=1, seed=21)
    A.append(UPDATE, I, J, V)
    E = len(I)
    c0, c1 = esp.owner_ranges(N, P)[self_rank]
    d.ck(d.lib.esp_set_column_window(d.h, c0 + 1, c1))
    ok = C.c_int32()
    pk, pv, pc = C.c_void_p(), C.c_void_p(), C.c_void_p()
    eoff = np.zeros(P + 1, np.int64)
    nb = C.c_int64()
    # (entries_per_shard is the planning hint every rank passes alike: 10^6 makes the plan 512 digits per shard at a size the
    # oracle follows -- the partitioned exchange declines problems below 2^8 segments per shard)
    d.ck(d.lib.esp_shard_partition(d.h, P, self_rank, 1000000, C.byref(ok), C.byref(pk), C.byref(pv), C.byref(pc),
                                   eoff.ctypes.data_as(C.c_void_p), C.byref(nb)))
    assert ok.value == 1, "the plan declined: the case does not reach esp_shard_assemble"
    own = (J > c0) & (J <= c1)
    assert eoff[self_rank + 1] - eoff[self_rank] == np.count_nonzero(own) and eoff[P] == E
    if self_rank > 0:
        assert eoff[self_rank] > 0        # (the start of the own range is not zero: the word that was shared)
    zeros = torch.zeros(int(nb.value), dtype=torch.int64, device="cuda")
    keys = (C.c_void_p * P)()
    vals = (C.c_void_p * P)()
    cnts = (C.c_void_p * P)(*[zeros.data_ptr()] * P)
    ne = np.zeros(P, np.int64)
    d.ck(d.lib.esp_shard_assemble(d.h, keys, vals, cnts, ne.ctypes.data_as(C.c_void_p), C.byref(ok)))
    assert ok.value == 1
    A.flush()
    assert A.debug_last_partition() == 7
    O = orc.ExtendableSparseMatrix(N, N)
    O.apply(np.full(np.count_nonzero(own), UPDATE, np.uint8), I[own], J[own], V[own])
    O.flush()
    assert_csc_equal(hip_arrays(A), O.arrays())


# ------------------------------------------------------------------ round 5: the field contract of the north-star type
def test_cscmatrix_field_contract_and_host_edits(esp, orc):
    """`A.cscmatrix` right after `flush!` (how the reference's consumers read it: factorizations/ilu0.jl:126-136,
    umfpack_lu.jl:18-27, jacobi.jl:54-64) is a valid host CSC: nothing travels when it is current, nzval only -- INTO the array
    handed out before -- when no position was added, the whole matrix otherwise; host edits of nonzeros (sprand.jl:82,
    test_parallel.jl:71-92) go back to the device in front of the next update.  The Python class mirrors
    HIPResidentSparseMatrixCSC of ESparseHIP.jl (host_csc!, push_edits!, touch!) call for call."""
    rng = np.random.default_rng(55)
    N = 400
    A = esp.ExtendableSparseMatrix(N, N)
    O = orc.ExtendableSparseMatrix(N, N)

    def batch(cnt, lo=1, hi=N):
        I, J, V = rng.integers(lo, hi + 1, cnt), rng.integers(lo, hi + 1, cnt), rng.standard_normal(cnt)
        return I, J, V

    def both(I, J, V, kind=UPDATE):
        A.append(kind, I, J, V)
        O.apply(np.full(len(I), kind, np.uint8), I, J, V)

    I1, J1, V1 = batch(5000, 1, N // 2)
    both(I1, J1, V1)
    A.flush(), O.flush()
    ph0 = A.phash
    c = A.cscmatrix                                   # STALE -> esp_get_csc
    check_julia_invariants(N, N, *c.arrays())
    assert_csc_equal(c.arrays(), O.arrays(), "field after flush!")
    assert A.cscmatrix is c and A.sparse() is c       # CURRENT: the same object, nothing travels
    nz_id = c.nzval
    c.nzval[:] = 0.0                                  # nonzeros(A) .= 0 on the host ...
    O.zero_values()
    both(I1, J1, 2.0 * V1)                            # ... then a re-assembly: every update hits a stored position
    A.flush(), O.flush()
    assert A.phash == ph0                             # no position added: phash kept (extendable.jl:249-252)
    c2 = A.cscmatrix                                  # VALUES_STALE -> esp_get_nzval into the array handed out before
    assert c2 is c and c2.nzval is nz_id
    assert_csc_equal(c2.arrays(), O.arrays(), "host edit + hits")
    c.nzval[::3] = 1.25                               # an edit with NO update behind it still reaches the device consumers
    cp_, rv_, nz_ = O.arrays()
    dense = np.zeros(N)
    x = rng.standard_normal(N)
    nzo = nz_.copy()
    nzo[::3] = 1.25
    for j in range(N):
        for k in range(cp_[j] - 1, cp_[j + 1] - 1):
            dense[rv_[k] - 1] += nzo[k] * x[j]
    r = A.mul(x)
    assert np.allclose(r, dense, rtol=1e-13, atol=1e-13)
    I2, J2, V2 = batch(3000)                          # new positions: the CSC is rebuilt
    O2 = orc.ExtendableSparseMatrix(N, N)             # (the oracle has no host-edit call: replay its matrix with the edited values)
    cols = np.repeat(np.arange(1, N + 1), np.diff(cp_))
    O2.apply(np.full(len(rv_), RAW, np.uint8), rv_, cols, nzo)
    O2.flush()
    A.append(UPDATE, I2, J2, V2)
    O2.apply(np.full(len(I2), UPDATE, np.uint8), I2, J2, V2)
    A.flush(), O2.flush()
    assert A.phash != ph0
    c3 = A.cscmatrix
    assert c3 is not c and c3.colptr is not c.colptr
    assert_csc_equal(c3.arrays(), O2.arrays(), "field after a flush! that added positions")
    # A.cscmatrix = B attaches B (reset! of the reference assigns the field)
    B = esp.ExtendableSparseMatrix(N, N, host_edits=False)
    B.cscmatrix = esp.SparseMatrixCSC(N, N, c3.colptr.copy(), c3.rowval.copy(), c3.nzval.copy())
    assert B.nnz() == c3.nnz()
    b = B.cscmatrix
    b.nzval[:] = 7.0                                  # host_edits = False: the caller promised not to, nothing goes up
    B.rawupdateindex("+", 0.5, int(I2[0]), int(J2[0]))
    A.rawupdateindex("+", 0.5, int(I2[0]), int(J2[0]))
    assert_csc_equal(B.cscmatrix.arrays(), A.cscmatrix.arrays(), "host_edits = False")


def test_int32_csc_transfers(esp, orc):
    """esp_get_csc_i32 / esp_set_csc_i32 (ExtendableSparseMatrix{Float64,Int32}: extendable.jl:10-25 is generic in Ti): the same
    matrix as the Int64 transfers, index arrays narrowed / widened on the device."""
    import ctypes as C
    n = 17
    A = esp.fdrand(n, n, n, rand_mode=1, seed=0x5EED0002)
    cp, rv, nz = A.sparse().arrays()
    d = A._d
    cp32, rv32, nz2 = np.full(len(cp), -1, np.int32), np.full(len(rv), -1, np.int32), np.zeros(len(nz))
    d.ck(d.lib.esp_get_csc_i32(d.h, C.c_void_p(cp32.ctypes.data), C.c_void_p(rv32.ctypes.data), C.c_void_p(nz2.ctypes.data)))
    assert np.array_equal(cp32, cp) and np.array_equal(rv32, rv) and np.array_equal(bits(nz2), bits(nz))
    B = esp.ExtendableSparseMatrix(n ** 3, n ** 3)
    b = B._d
    b.ck(b.lib.esp_set_csc_i32(b.h, C.c_void_p(cp32.ctypes.data), C.c_void_p(rv32.ctypes.data), C.c_void_p(nz2.ctypes.data), len(rv)))
    B._phash = None
    assert_csc_equal(B._d.get_csc().arrays(), (cp, rv, nz))
    rng = np.random.default_rng(1)
    I, J, V = rng.integers(1, n ** 3 + 1, 4000), rng.integers(1, n ** 3 + 1, 4000), rng.standard_normal(4000)
    O = orc.fdrand(n, n, n, rand_mode=1, seed=0x5EED0002, style=orc.KIND_UPDATE)
    B.append(UPDATE, I.astype(np.int32), J.astype(np.int32), V)        # esp_append_host_i32
    O.apply(np.full(4000, UPDATE, np.uint8), I, J, V)
    assert_csc_equal(B.sparse().arrays(), O.arrays())
    empty = esp.ExtendableSparseMatrix(5, 5)
    e = empty._d
    cpe = np.zeros(6, np.int32)
    e.ck(e.lib.esp_get_csc_i32(e.h, C.c_void_p(cpe.ctypes.data), None, None))
    assert np.array_equal(cpe, np.ones(6, np.int32))


def test_elements_plan_is_superseded_by_any_later_element_append(esp, orc):
    """a kept plan is that of the LAST esp_append_elements.  A later element-level append that does not
    take the item partition (here: a non-empty buffer; also a repeated node, a BoundsError) must not leave the plan of the
    earlier mesh behind: esp_append_elements_again then returns ESP_ERR_STATE instead of assembling mesh A's connectivity
    with mesh B's element matrices."""
    import torch
    dim, npd = 2, 120
    nn, nloc = npd ** dim, dim + 1
    cnA, emA, dgA = orc.fem_mesh(dim, npd, seed=0x5EED0004, order_mode=1, node_mode=0)
    cnB, emB, dgB = orc.fem_mesh(dim, npd, seed=0x5EED0104, order_mode=1, node_mode=1)
    P = esp.ExtendableSparseMatrix(nn, nn)
    P.elements_keep_plan()
    P.append_elements(cnA, emA, dgA)
    P.flush()
    P.append_elements_again(np.asfortranarray(emA * 2.0), np.asfortranarray(dgA))       # the plan of mesh A works
    P.flush()
    P.rawupdateindex("+", 1.0, 1, 1)
    P.append_elements(cnB, emB, dgB)                  # non-empty buffer: stream order, no plan is made ...
    P.flush()
    with pytest.raises(esp.EspError) as ei:
        P.append_elements_again(np.asfortranarray(emB), np.asfortranarray(dgB))          # ... and mesh A's must be gone
    assert ei.value.code == esp._lib.ESP_ERR_STATE
    assert P.nnznew() == 0
    bad = cnA.copy()
    bad[1, 7] = nn + 1
    P.reset()
    P.append_elements(cnA, emA, dgA)                  # a planned call again ...
    P.flush()
    with pytest.raises(esp.BoundsError):
        P.append_elements(bad, emA, dgA)              # ... superseded by a call that fails
    with pytest.raises(esp.EspError):
        P.append_elements_again(np.asfortranarray(emA), np.asfortranarray(dgA))


def test_flush_sum_failure_leaves_everything_clean(esp, orc):
    """esp_flush_sum is all-or-nothing.  A buffer whose own flush fails (here: a declared column window with an entry
    outside it -> ESP_ERR_STATE) makes the call fail; every buffer still comes back EMPTY (nothing pending, no matrix of its
    own), the destination keeps its stored matrix and has nothing pending, and the next esp_flush_sum works."""
    import ctypes as C
    m = n = 600
    rng = np.random.default_rng(9)
    home = esp.SparseMatrixHIPCOO(m, n)
    xs = [esp.SparseMatrixHIPCOO(m, n) for _ in range(3)]
    I, J, V = rng.integers(1, m + 1, 3000), rng.integers(1, n + 1, 3000), rng.standard_normal(3000)
    csc = esp.SparseMatrixCSC(m, n)
    xs[0].append(UPDATE, I, J, V)
    first = esp.SparseMatrixHIPCOO.sum(xs, csc, home=home)
    O = orc.ExtendableSparseMatrix(m, n)
    O.apply(np.full(3000, UPDATE, np.uint8), I, J, V)
    assert_csc_equal(first.arrays(), O.arrays())
    # round 2: buffer 1 promises columns 1..100 and breaks the promise
    xs[0].append(UPDATE, I, J, V)
    d1 = xs[1]._d
    d1.ck(d1.lib.esp_set_column_window(d1.h, 1, 100))
    xs[1].append(UPDATE, np.array([5, 6]), np.array([7, 450]), np.array([1.0, 2.0]))
    xs[2].append(UPDATE, I[:10], J[:10], V[:10])
    with pytest.raises(esp.EspError):
        esp.SparseMatrixHIPCOO.sum(xs, first, home=home)
    for x in xs:
        assert x.nnz() == 0 and x._d.nnz() == 0
    assert home._d.pending() == 0 and home._d.nnz() == first.nnz()
    assert_csc_equal(home._d.get_csc().arrays(), O.arrays(), "the destination after a failed sum")
    d1.ck(d1.lib.esp_set_column_window(d1.h, 1, n))
    xs[1].append(UPDATE, I, J, V)
    out = esp.SparseMatrixHIPCOO.sum(xs, first, home=home)
    L = orc.SparseMatrixLNK(m, n)
    for i, j, v in zip(I.tolist(), J.tolist(), V.tolist()):
        L.updateindex(orc.OP_ADD, v, i, j)
    assert_csc_equal(out.arrays(), (L + orc.CSC(m, n, *O.arrays())).arrays(), "the sum after the failed one")


def test_elements_from_arrays_at_odd_offsets(esp, orc):
    """Device arrays that are 8- but not 16-byte aligned (views that start one element into a larger array): the cells kernel
    takes its 8-byte loads instead of the 16-byte ones; aligned arrays of the same mesh.  3-D (4 nodes per cell) and 2-D."""
    import torch
    for dim, npd in ((3, 22), (2, 150)):
        nn = npd ** dim
        cn, em, dg = orc.fem_mesh(dim, npd, seed=0x5EED0004, order_mode=1, node_mode=0)
        I, J, V = orc.elements_stream(cn, em, dg)
        O = orc.ExtendableSparseMatrix(nn, nn)
        O.apply(np.full(len(I), RAW, np.uint8), I, J, V)
        O.flush()
        flat = lambda x: np.ascontiguousarray(np.asfortranarray(x).ravel(order="F"))   # noqa: E731  (Julia's layout, flat)
        for off in (1, 0):
            dev = []
            for x in (flat(cn), flat(em), flat(dg)):
                buf = torch.zeros(len(x) + 2, dtype=torch.from_numpy(x).dtype, device="cuda")
                buf[off:off + len(x)] = torch.from_numpy(x).cuda()
                dev.append(buf[off:off + len(x)])
            torch.cuda.synchronize()
            assert all(t.data_ptr() % 16 == 8 * off for t in dev)
            A = esp.ExtendableSparseMatrix(nn, nn)
            A.append_elements(dev[0], dev[1], dev[2])
            A.flush()
            assert A.debug_last_lazy_items() == 1
            assert_csc_equal(hip_arrays(A), O.arrays(), "dim %d offset %d" % (dim, off))


def test_lazy_item_batches_and_everything_that_expands_them(esp, orc):
    """A batch of an item partition stays a list of sorted items until its flush (group3_items.hpp: the bucket kernel forms the
    updates itself).  Whatever else touches the pending entries first must find them as entries: an append behind the batch, a
    clone, getindex on the buffer, nnznew, a flush over a stored pattern (the second assembly), reset!, a second batch right
    behind the first, esp_flush_sum.  Every result against the oracle fed the same calls one by one; op '-', updateindex! with
    zeros, meshes without a diagonal term; 2-D and 3-D."""
    import torch
    rng = np.random.default_rng(77)
    for dim, npd in ((2, 220), (3, 26)):
        nn, nloc = npd ** dim, dim + 1
        cn, em, dg = orc.fem_mesh(dim, npd, seed=0x5EED0004, order_mode=1, node_mode=0)
        em[:, :, ::7] = 0.0                                 # (zeros: updateindex! creates nothing for them)
        I, J, V = orc.elements_stream(cn, em, dg)
        In, Jn, Vn = orc.elements_stream(cn, em, None)

        def oracle(calls):
            O = orc.ExtendableSparseMatrix(nn, nn)
            for kind, (i, j, v) in calls:
                O.apply(np.full(len(i), kind, np.uint8), i, j, v)
            O.flush()
            return O.arrays()

        # plain: the fused kernel, every kind / op it takes, with and without the diagonal term
        for kind, op, diag in ((RAW, "+", True), (UPDATE, "+", True), (RAW, "-", True), (UPDATE, "-", False), (RAW, "+", False)):
            A = esp.ExtendableSparseMatrix(nn, nn)
            A.append_elements(cn, em, dg if diag else None, kind=kind, op=op)
            A.flush()
            assert A.debug_last_lazy_items() == 1 and A.debug_last_partition() == 4, (kind, op, diag)
            i, j, v = (I, J, V) if diag else (In, Jn, Vn)
            assert_csc_equal(hip_arrays(A), oracle([(kind, (i, j, v if op == "+" else -v))]), "plain %d %s %s" % (kind, op, diag))
        # SET is not the fused kernel's: expanded at append time
        A = esp.ExtendableSparseMatrix(nn, nn)
        A.append_elements(cn, em, dg, kind=SET)
        A.flush()
        assert A.debug_last_lazy_items() == 0
        assert_csc_equal(hip_arrays(A), oracle([(SET, (I, J, V))]), "SET")
        # an append behind the batch (then the batch + tail flush), a second batch behind the first
        I2, J2, V2 = rng.integers(1, nn + 1, 5000), rng.integers(1, nn + 1, 5000), rng.standard_normal(5000)
        A = esp.ExtendableSparseMatrix(nn, nn)
        A.append_elements(cn, em, dg)
        A.append(UPDATE, I2, J2, V2)
        A.flush()
        assert A.debug_last_lazy_items() == 0
        assert_csc_equal(hip_arrays(A), oracle([(RAW, (I, J, V)), (UPDATE, (I2, J2, V2))]), "tail")
        A.reset()
        A.append_elements(cn, em, dg)
        A.append_elements(cn, em, dg)
        assert A.nnznew() == 2 * len(I)
        A.flush()
        assert_csc_equal(hip_arrays(A), oracle([(RAW, (I, J, V)), (RAW, (I, J, V))]), "twice")
        # (a handle that met column runs above 128 -- "twice" in 3-D -- has learnt that group3_k is not for it: no items any more)
        A.append_elements(cn, em, dg)
        A.flush()
        assert_csc_equal(hip_arrays(A), oracle([(RAW, (I, J, V))] * 3), "over the stored pattern, after twice")
        # the second assembly runs over the stored pattern (every update hits): items again, the fused kernel's re-assembly form
        A = esp.ExtendableSparseMatrix(nn, nn)
        A.append_elements(cn, em, dg)
        A.flush()
        A.append_elements(cn, em, dg)
        A.flush()
        assert A.debug_last_lazy_items() == 1 and A.debug_last_local_small() == 4, (A.debug_last_lazy_items(), A.debug_last_local_small())
        assert_csc_equal(hip_arrays(A), oracle([(RAW, (I, J, V))] * 2), "over the stored pattern")
        # ... a third one, then one that brings NEW positions (the re-assembly form refuses: expanded, the general kernels)
        A.append_elements(cn, em, dg)
        A.flush()
        assert A.debug_last_lazy_items() == 1 and A.debug_last_local_small() == 4, (A.debug_last_lazy_items(), A.debug_last_local_small())
        assert_csc_equal(hip_arrays(A), oracle([(RAW, (I, J, V))] * 3), "re-assembly, fused")
        # (the same mesh with shifted node numbers: mostly positions the pattern lacks)
        A.append_elements(np.asfortranarray((cn + 7 - 1) % nn + 1), em, dg)
        A.flush()
        Is, Js, Vs = orc.elements_stream(np.asfortranarray((cn + 7 - 1) % nn + 1), em, dg)
        assert_csc_equal(hip_arrays(A), oracle([(RAW, (I, J, V))] * 3 + [(RAW, (Is, Js, Vs))]), "new positions over the pattern")
        # a time step through the kept plan (esp_append_elements_again): the same
        P = esp.ExtendableSparseMatrix(nn, nn)
        P.elements_keep_plan()
        P.append_elements(cn, em, dg)
        P.flush()
        calls = [(RAW, (I, J, V))]
        for step in range(3):
            em2, dg2 = np.asfortranarray(em * (1.5 + step)), np.asfortranarray(dg * 0.5)
            P.append_elements_again(em2, dg2)
            P.flush()
            calls.append((RAW, orc.elements_stream(cn, em2, dg2)))
            assert P.debug_last_lazy_items() == 1, (step, P.debug_last_lazy_items())
            assert_csc_equal(hip_arrays(P), oracle(calls), "time step %d" % step)
        # clone of a handle with a lazy batch: both flush to the same matrix; getindex on the pending buffer
        A = esp.ExtendableSparseMatrix(nn, nn)
        A.append_elements(cn, em, dg)
        B = A.copy()
        want = oracle([(RAW, (I, J, V))])
        assert_csc_equal(hip_arrays(B), want, "clone")
        assert_csc_equal(hip_arrays(A), want, "the cloned handle")
        X = esp.SparseMatrixHIPCOO(nn, nn)
        X.append_elements(cn, em, dg)
        i0, j0 = int(I[5]), int(J[5])
        ref = orc.SparseMatrixLNK(nn, nn)
        for i, j, v in zip(I.tolist(), J.tolist(), V.tolist()):
            if j == j0 and i == i0:
                ref.rawupdateindex(orc.OP_ADD, v, i, j)
        assert bits(np.array([X[i0, j0]])) == bits(np.array([ref[i0, j0]]))
        csc = X + esp.SparseMatrixCSC(nn, nn)
        assert_csc_equal(csc.arrays(), want, "buffer + csc after getindex")
        # the generator's items the same way: tail behind the batch, reset with a batch pending, then a clean assembly
        G = esp.ExtendableSparseMatrix(nn, nn)
        G.generate_fem(dim, npd, seed=0x5EED0004, order_mode=1)
        G.append(UPDATE, I2, J2, V2)
        G.flush()
        Ig, Jg, Vg = orc.fem_stream(dim, npd, seed=0x5EED0004, order_mode=1)
        assert_csc_equal(hip_arrays(G), oracle([(RAW, (Ig, Jg, Vg)), (UPDATE, (I2, J2, V2))]), "generator + tail")
        G.reset()
        G.generate_fem(dim, npd, seed=0x5EED0004, order_mode=1)
        G.reset()
        G.generate_fem(dim, npd, seed=0x5EED0009, order_mode=1)
        G.flush()
        assert G.debug_last_lazy_items() == 1
        Ig, Jg, Vg = orc.fem_stream(dim, npd, seed=0x5EED0009, order_mode=1)
        assert_csc_equal(hip_arrays(G), oracle([(RAW, (Ig, Jg, Vg))]), "generator after reset")


def test_generator_repeats_its_plan(esp, orc):
    """esp_generate_fdrand on the same grid again (a time loop: reset!, fdrand!, flush!): the second call finds the run lists
    and bucket starts of the first -- a function of the grid and the plan, not of seed, values or kind of randomness -- and goes
    straight to its PART launch (esp_debug_last_plan_reused).  Another grid, another kind, an append in between, a partition
    of another stream in between, force_path 31: the full path.  Always the oracle's bits."""
    import torch
    for (nx, ny, nz) in ((48, 48, 48), (300, 201, 5)):
        N = nx * ny * nz
        A = esp.ExtendableSparseMatrix(N, N)

        def run(seed, kind, expect, what, mode=1):
            A.reset()
            A.generate_fdrand(nx, ny, nz, seed=seed, rand_mode=mode, kind=kind)
            assert A.debug_last_plan_reused() == expect, (what, A.debug_last_plan_reused())
            A.flush()
            assert A.debug_last_partition() == 4
            O = orc.fdrand(nx, ny, nz, rand_mode=mode, seed=seed, style=kind)
            assert_csc_equal(hip_arrays(A), O.arrays(), what)

        run(1, UPDATE, 0, "first")
        run(2, UPDATE, 1, "same grid, new values")
        run(3, UPDATE, 1, "again", mode=2)
        run(4, RAW, 0, "another kind")
        run(5, RAW, 1, "that kind again")
        # an append of another stream in between rewrites the tables: the full path, then reuse again
        A.reset()
        I, J, V = orc.fdrand_stream(nx, ny, nz, rand_mode=1, seed=9)
        dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()   # noqa: E731
        A.append_device(UPDATE, dev(I[::2]), dev(J[::2]), dev(V[::2]))
        A.flush()
        run(6, RAW, 0, "after a partition of another stream")
        run(7, RAW, 1, "and again")
        # behind pending entries the generator does not partition at all; afterwards the plan still stands
        A.reset()
        A.rawupdateindex("+", 1.0, 1, 1)
        A.generate_fdrand(nx, ny, nz, seed=8, rand_mode=1, kind=RAW)
        assert A.debug_last_plan_reused() == 0
        A.flush()
        O = orc.ExtendableSparseMatrix(N, N)
        O.rawupdateindex(orc.OP_ADD, 1.0, 1, 1)
        I, J, V = orc.fdrand_stream(nx, ny, nz, rand_mode=1, seed=8)
        O.apply(np.full(len(I), RAW, np.uint8), I, J, V)
        O.flush()
        assert_csc_equal(hip_arrays(A), O.arrays(), "behind a pending entry")
        B = esp.ExtendableSparseMatrix(N, N)
        B.debug_force_path(31)
        for seed in (1, 2):
            B.reset()
            B.generate_fdrand(nx, ny, nz, seed=seed, rand_mode=1, kind=UPDATE)
            assert B.debug_last_plan_reused() == 0
            B.flush()
        assert_csc_equal(hip_arrays(B), orc.fdrand(nx, ny, nz, rand_mode=1, seed=2, style=orc.KIND_UPDATE).arrays(), "force 31")


def test_flush_sum_over_element_batches(esp, orc):
    """Base.sum(xmatrices, csc) when every buffer was filled by esp_append_elements (a mesh dealt to the buffers: bands of the cell
    list, or cells dealt round-robin): the folds of all buffers are ONE launch of the fused bucket kernel over their item records,
    the combine flush reads the folded records as pieces (esp_debug_last_lazy_items(home) == 2).  Small mesh: against the oracle's
    MT wrapper fed every call with its tid; larger meshes: bit for bit the per-buffer path (force_path 39 on the buffers), fresh
    and over the stored pattern, with empty buffers, with and without the diagonal term, '-' in one buffer."""
    import ctypes as C

    def lazy_state(home):
        v = C.c_int32()
        home._d.ck(home._d.lib.esp_debug_last_lazy_items(home._d.h, C.byref(v)))
        return v.value

    def join_state(home):
        v = C.c_int32()
        home._d.ck(home._d.lib.esp_debug_last_sum_join(home._d.h, C.byref(v)))
        return v.value

    def deal(cn, em, dg, p, how):
        nc = cn.shape[1]
        if how == "bands":
            idx = [np.arange(nc * t // p, nc * (t + 1) // p) for t in range(p)]
        else:
            idx = [np.arange(t, nc, p) for t in range(p)]
        return [(np.asfortranarray(cn[:, i]), np.asfortranarray(em[:, :, i]), None if dg is None else np.asfortranarray(dg[:, i])) for i in idx]

    # small: the oracle's MT wrapper (sparse! over csc, buffer 1, buffer 2, ...)
    dim, npd, p = 2, 60, 4
    nn = npd ** dim
    cn, em, dg = orc.fem_mesh(dim, npd, seed=0x5EED0004, order_mode=0, node_mode=0)
    parts = deal(cn, em, dg, p, "bands")
    xs = [esp.SparseMatrixHIPCOO(nn, nn) for _ in range(p)]
    home = esp.SparseMatrixHIPCOO(nn, nn)
    O = orc.CSC(nn, nn)
    csc = esp.SparseMatrixCSC(nn, nn)
    for rnd in range(2):                       # fresh, then over the stored pattern: sparse!(I, J, V) over (csc, buffer 1, buffer 2, ...)
        lnks = []
        for t, (c, e, d) in enumerate(parts):
            xs[t].append_elements(c, e, d)
            I, J, V = orc.elements_stream(c, e, d)
            L = orc.SparseMatrixLNK(nn, nn)
            for i, j, v in zip(I.tolist(), J.tolist(), V.tolist()):
                L.rawupdateindex(orc.OP_ADD, v, i, j)
            lnks.append(L)
        csc = esp.SparseMatrixHIPCOO.sum(xs, csc, home=home)
        assert lazy_state(home) == 2, (rnd, lazy_state(home))
        for L in lnks:
            O = L + O                          # (successive csc + buffer: each buffer folded by itself, sparsematrixdilnkc.jl:397-435)
        assert_csc_equal(csc.arrays(), O.arrays(), "round %d" % rnd)
    # larger: device against device (the per-buffer folds: force_path 39 keeps every batch expanded)
    joins = []     # (segments of the folds' plan the combine flush joined: bands of a mesh leave a fraction of a segment each)
    for dim, npd, p, how, diag, neg in ((2, 400, 16, "bands", True, -1), (2, 300, 5, "round_robin", False, 2), (3, 30, 8, "bands", True, 0)):
        nn = npd ** dim
        cn, em, dg = orc.fem_mesh(dim, npd, seed=0x5EED0004, order_mode=0, node_mode=0)
        parts = deal(cn, em, dg if diag else None, p, how)
        results = []
        for force in (0, 39):
            xs = [esp.SparseMatrixHIPCOO(nn, nn) for _ in range(p)]
            for x in xs:
                x._d.ck(x._d.lib.esp_debug_force_path(x._d.h, force))
            home = esp.SparseMatrixHIPCOO(nn, nn)
            csc = esp.SparseMatrixCSC(nn, nn)
            for rnd in range(2):
                for t, (c, e, d) in enumerate(parts):
                    if t == 1 and rnd == 0:
                        continue                                       # an empty buffer
                    xs[t].append_elements(c, e, d, op="-" if t == neg else "+")
                csc = esp.SparseMatrixHIPCOO.sum(xs, csc, home=home)
                if force == 0 and how == "bands":
                    assert lazy_state(home) == 2, (dim, npd, p, how, rnd, lazy_state(home))
                    joins.append(join_state(home))
                elif force == 0:
                    # (cells dealt round-robin: every buffer covers every column, a segment's records of all buffers together exceed
                    # the combine kernel's capacity -- the joint path declines after its folds and the general path takes the buffers
                    # as they are: the same result)
                    assert lazy_state(home) in (0, 2)
                else:
                    assert lazy_state(home) != 2
                results.append((force, rnd, csc.arrays()))
        for rnd in range(2):
            a = [r[2] for r in results if r[0] == 0 and r[1] == rnd][0]
            b = [r[2] for r in results if r[0] == 39 and r[1] == rnd][0]
            assert_csc_equal(a, b, "%d-D %d buffers %s round %d" % (dim, p, how, rnd))
    assert max(joins) > 1 and all(j in (1, 2, 4, 8) for j in joins), joins
    # buffers whose own plans differ (every handle plans its item partition from its own history; here: the plan hook on two of
    # them, one / two prefix bits more): the joint path takes the coarsest plan and reads the finer buffers' tables at it
    dim, npd, p = 2, 400, 8
    nn = npd ** dim
    cn, em, dg = orc.fem_mesh(dim, npd, seed=0x5EED0004, order_mode=0, node_mode=0)
    parts = deal(cn, em, dg, p, "bands")
    results = []
    for caps in ((0, 0, 0, 0, 0, 0, 0, 0), (0, 0, 0, 400.0, 0, 0, 200.0, 0)):
        xs = [esp.SparseMatrixHIPCOO(nn, nn) for _ in range(p)]
        for x, cap in zip(xs, caps):
            x._d.ck(x._d.lib.esp_debug_plan_cap(x._d.h, C.c_double(cap)))
        home = esp.SparseMatrixHIPCOO(nn, nn)
        csc = esp.SparseMatrixCSC(nn, nn)
        for rnd in range(2):
            for t, (c, e, d) in enumerate(parts):
                xs[t].append_elements(c, e, d)
            csc = esp.SparseMatrixHIPCOO.sum(xs, csc, home=home)
            assert lazy_state(home) == 2, (caps, rnd, lazy_state(home))
            lo, hi = C.c_int32(), C.c_int32()
            home._d.ck(home._d.lib.esp_debug_last_sum_plan_bits(home._d.h, C.byref(lo), C.byref(hi)))
            assert (hi.value > lo.value) == any(caps), (caps, lo.value, hi.value)     # (the plans really differ)
            results.append(csc.arrays())
    assert_csc_equal(results[0], results[2], "plans that differ, fresh")
    assert_csc_equal(results[1], results[3], "plans that differ, stored pattern")
