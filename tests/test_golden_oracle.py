"""CPU: the oracle reproduces every committed golden fixture (regression pin of tests/golden/)."""
import numpy as np
import pytest

import golden_util as gu
from refmodel import assert_csc_equal, check_julia_invariants


@pytest.mark.parametrize("name", gu.STREAMS)
def test_stream_fixture(orc, name):
    fx = gu.load(name)
    for q, got, want, _ in gu.replay(fx, orc.ExtendableSparseMatrix,
                                     lambda A, k, I, J, V: A.apply(k, I, J, V),
                                     lambda A: A.flush(), lambda A: A.arrays()):
        assert_csc_equal(got, want, "%s flush %d" % (name, q))
        check_julia_invariants(int(fx["m"]), int(fx["n"]), *got)


def test_updates_trace_nnz(orc):
    fx = gu.load("updates_trace")
    assert [len(fx["rowval%d" % q]) for q in range(3)] == [2, 3, 3]   # test_updates.jl:16,18,22


@pytest.mark.parametrize("dims", [(100, 1, 1), (10, 10, 1), (5, 5, 5)])
@pytest.mark.parametrize("mode", [0, 1])
def test_fdrand_small(orc, dims, mode):
    fx = gu.load("fdrand_small")
    tag = "fd_%dx%dx%d_m%d" % (*dims, mode)
    got = orc.fdrand(*dims, rand_mode=mode, seed=0x5EED0002, style=orc.KIND_RAWUPDATE).arrays()
    assert_csc_equal(got, (fx[tag + "_colptr"], fx[tag + "_rowval"], fx[tag + "_nzval"]), tag)


@pytest.mark.parametrize("mode", [0, 1])
def test_fdrand_30_digest(orc, mode):
    d = gu.digests()["fd_30x30x30_m%d" % mode]
    arrs = orc.fdrand(30, 30, 30, rand_mode=mode, seed=0x5EED0002, style=orc.KIND_UPDATE).arrays()
    assert len(arrs[1]) == int(d["nnz"]) == 183600
    assert gu.digest(*arrs) == d["csc"]
    assert gu.digest(*orc.fdrand_stream(30, 30, 30, rand_mode=mode, seed=0x5EED0002)) == d["stream"]


@pytest.mark.parametrize("dim,npd", [(2, 32), (3, 10)])
def test_fem_small(orc, dim, npd):
    fx = gu.load("fem_small")
    tag = "fem%dd_%d" % (dim, npd)
    nn, nc, cnt = orc.fem_sizes(dim, npd)
    I, J, V = orc.fem_stream(dim, npd, seed=0x5EED0004, order_mode=1)
    assert gu.digest(I, J, V) == gu.digests()[tag]["stream"]
    A = orc.ExtendableSparseMatrix(nn, nn)
    A.apply(np.full(cnt, 2, np.uint8), I, J, V)
    assert_csc_equal(A.arrays(), (fx[tag + "_colptr"], fx[tag + "_rowval"], fx[tag + "_nzval"]), tag)
