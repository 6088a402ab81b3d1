"""fdrand!/fdrand of the reference (src/matrix/sprand.jl:58-126,226-256) on top of the mirror classes.

`fdrand_` walks the reference's triple loop on the host and issues the same update calls (any
matrix class, any update style) -- the form the reference's tests use at small sizes.
`fdrand` with device=True produces the identical update stream with the on-device generator
(esp_generate_fdrand) and is what the benchmark times.
"""
import numpy as np

from ._lib import ESP_COO, ESP_RAWUPDATE, ESP_UPDATE
from .matrix import ExtendableSparseMatrix

MASK = (1 << 64) - 1


def _mix64(z):
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
    return z ^ (z >> 31)


def uniform(seed, counter):
    """Counter-based U[0,1): same formula as the device generator (csrc/common.hpp:esp_uniform)."""
    z = _mix64((seed + (counter + 1) * 0x9E3779B97F4A7C15) & MASK)
    return (z >> 11) * 2.0 ** -53


def make_rand(rand_mode, seed):
    if rand_mode == 0:
        return lambda ctr: 1.0
    if rand_mode == 1:
        return lambda ctr: 0.1 + uniform(seed, ctr)
    return lambda ctr: uniform(seed, ctr)


def update_pluseq(A, v, i, j):       # update = (A,v,i,j)->A[i,j]+=v   (sprand.jl:62)
    A[i, j] = A[i, j] + v


def update_updateindex(A, v, i, j):  # docs/src/example.md:155-157
    A.updateindex("+", v, i, j)


def update_rawupdateindex(A, v, i, j):
    A.rawupdateindex("+", v, i, j)


def stencil_updates(nx, ny, nz, rand, update):
    """The update stream of fdrand! (sprand.jl:87-124) as calls update(v, i, j), in the reference's order: per node l
    (x fastest) the x-pair if i < nx, the x-boundary term, the y-pair, the y-boundary term (ny > 2), the z-pair, the
    z-boundary term (nz > 2); a pair is (-v,l,l'), (-v,l',l), (v,l,l), (v,l',l') (:87-92); draw k of node l has the
    counter 6 (l - 1) + k.  The one host-side statement of the stream: fdrand_ and fdrand_coo both use it."""
    def update_pair(v, i, j):
        update(-v, i, j)
        update(-v, j, i)
        update(v, i, i)
        update(v, j, j)

    hx, hy, hz = 1.0 / nx, 1.0 / ny, 1.0 / nz
    nxy = nx * ny
    l = 1
    for k in range(1, nz + 1):
        for j in range(1, ny + 1):
            for i in range(1, nx + 1):
                c = 6 * (l - 1)
                if i < nx:
                    update_pair(rand(c + 0) * hy * hz / hx, l, l + 1)
                if i == 1 or i == nx:
                    update(rand(c + 1) * hy * hz, l, l)
                if j < ny:
                    update_pair(rand(c + 2) * hx * hz / hy, l, l + nx)
                if ny > 2 and (j == 1 or j == ny):
                    update(rand(c + 3) * hx * hz, l, l)
                if k < nz:
                    update_pair(rand(c + 4) * hx * hy / hz, l, l + nxy)
                if nz > 2 and (k == 1 or k == nz):
                    update(rand(c + 5) * hx * hy, l, l)
                l += 1


def fdrand_(A, nx, ny=1, nz=1, update=update_updateindex, rand_mode=2, seed=0x5EED0002):
    """fdrand!(A,nx,ny,nz;update,rand): sprand.jl:58-126."""
    N = nx * ny * nz
    if A.shape != (N, N):
        raise ValueError("Matrix size mismatch")
    A.zero_values()
    stencil_updates(nx, ny, nz, make_rand(rand_mode, seed), lambda v, i, j: update(A, v, i, j))
    A.flush()
    return A


def fdrand_device_(A, nx, ny=1, nz=1, rand_mode=2, seed=0x5EED0002, kind=ESP_UPDATE):
    """fdrand! with the hot loop generated on the GPU (same stream, same order, same bits)."""
    N = nx * ny * nz
    if A.shape != (N, N):
        raise ValueError("Matrix size mismatch")
    A.zero_values()
    A.generate_fdrand(nx, ny, nz, seed=seed, rand_mode=rand_mode, kind=kind)
    A.flush()
    return A


def fdrand(nx, ny=1, nz=1, rand_mode=1, seed=0x5EED0002, update=None, device=True, **kw):
    """fdrand(Float64,nx,ny,nz; matrixtype=ExtendableSparseMatrix): sprand.jl:226-256."""
    N = nx * ny * nz
    A = ExtendableSparseMatrix(N, N, **kw)
    if update is None and device:
        return fdrand_device_(A, nx, ny, nz, rand_mode, seed)
    return fdrand_(A, nx, ny, nz, update or update_updateindex, rand_mode, seed)


def fdrand_coo(nx, ny=1, nz=1, rand_mode=2, seed=0x5EED0002, device=True, **kw):
    """fdrand_coo(T,nx,ny,nz;rand) (sprand.jl:134-185): the stencil as COO triplets, then
    sparse(I,J,V).  device=True: the triplets are produced by the on-device generator as COO entries
    (same stream, same order); device=False: host triplets through ExtendableSparseMatrix.from_coo."""
    N = nx * ny * nz
    if device:
        A = ExtendableSparseMatrix(N, N, **kw)
        A.generate_fdrand(nx, ny, nz, seed=seed, rand_mode=rand_mode, kind=ESP_COO)
        A.flush()
        return A
    I, J, V = [], [], []

    def triplet(v, i, j):
        I.append(i)
        J.append(j)
        V.append(v)

    stencil_updates(nx, ny, nz, make_rand(rand_mode, seed), triplet)
    return ExtendableSparseMatrix.from_coo(I, J, V, N, N, **kw)
