"""Generates the committed golden fixtures under tests/golden/ from the CPU oracle.

Run:  python tests/golden/make_golden.py
The reference itself (Julia) cannot run in this image and stores no vectors,
so these vectors are outputs of oracle/esparse_oracle.c after it passed tests/test_oracle.py
(known-answer tests of the reference's suite + an independent dict/SciPy model).  Fixtures are data
only: update streams (kinds, I, J, V, flush positions) and the expected CSC after every flush.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

SET, UPDATE, RAW, PLUSEQ = 0, 1, 2, 3


def digest(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def stream_fixture(name, m, n, kinds, I, J, V, flush_after):
    """flush_after: sorted positions p meaning flush! after the first p updates."""
    A = orc.ExtendableSparseMatrix(m, n)
    out = {"m": m, "n": n, "kinds": kinds.astype(np.uint8), "I": I.astype(np.int64), "J": J.astype(np.int64),
           "V": V.astype(np.float64), "flush_after": np.array(flush_after, np.int64)}
    prev = 0
    for q, p in enumerate(flush_after):
        A.apply(kinds[prev:p], I[prev:p], J[prev:p], V[prev:p])
        rebuilt = A.flush()
        cp, rv, nz = A.arrays()
        out["colptr%d" % q], out["rowval%d" % q], out["nzval%d" % q] = cp, rv, nz
        out["rebuilt%d" % q] = np.array([int(rebuilt)])
        prev = p
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "updates", len(I), "final nnz", len(rv))


def main():
    rng = np.random.default_rng(20240807)
    # (1) test_updates.jl script trace
    kinds = np.array([SET, UPDATE, UPDATE, RAW, RAW], np.uint8)
    I = np.array([1, 4, 2, 2, 2])
    J = np.array([3, 5, 3, 3, 3])
    V = np.array([5.0, 6.0, 0.0, 0.0, 0.1])
    stream_fixture("updates_trace", 10, 10, kinds, I, J, V, [3, 4, 5])

    # (2) three test_assembly-style random streams with SET/UPDATE/RAW mixes, zeros, 2-3 splices
    pool = np.array([0.0, -0.0, 1.5, -1.5, 2.0 ** -1060, 3.25, -7.0])
    for name, m, n, cnt, nspl in (("assembly_a", 100, 100, 1500, 2), ("assembly_b", 200, 100, 3000, 3),
                                  ("assembly_c", 1000, 2000, 15000, 3)):
        kinds = rng.integers(0, 3, cnt).astype(np.uint8)
        I = rng.integers(1, m + 1, cnt)
        J = rng.integers(1, n + 1, cnt)
        # bias towards collisions so that SET/ADD order matters
        hot = rng.random(cnt) < 0.5
        I[hot] = rng.integers(1, min(m, 12) + 1, hot.sum())
        J[hot] = rng.integers(1, min(n, 9) + 1, hot.sum())
        V = np.where(rng.random(cnt) < 0.3, rng.choice(pool, cnt), 1.0 + rng.random(cnt))
        cuts = sorted(rng.choice(np.arange(1, cnt), nspl - 1, replace=False).tolist()) + [cnt]
        stream_fixture(name, m, n, kinds, I, J, V, cuts)

    # (3) stencils: full CSC for the small ones, digests for 30^3
    small = {}
    for dims in ((100, 1, 1), (10, 10, 1), (5, 5, 5)):
        for mode in (0, 1):
            cp, rv, nz = orc.fdrand(*dims, rand_mode=mode, seed=0x5EED0002, style=orc.KIND_UPDATE).arrays()
            tag = "fd_%dx%dx%d_m%d" % (*dims, mode)
            small[tag + "_colptr"], small[tag + "_rowval"], small[tag + "_nzval"] = cp, rv, nz
    np.savez_compressed(os.path.join(HERE, "fdrand_small.npz"), **small)
    lines = []
    for mode in (0, 1):
        cp, rv, nz = orc.fdrand(30, 30, 30, rand_mode=mode, seed=0x5EED0002, style=orc.KIND_UPDATE).arrays()
        I, J, V = orc.fdrand_stream(30, 30, 30, rand_mode=mode, seed=0x5EED0002)
        lines.append("fd_30x30x30_m%d nnz=%d csc=%s stream=%s" % (mode, len(rv), digest(cp, rv, nz), digest(I, J, V)))
    # (4) FEM meshes (~10^3 nodes)
    fem = {}
    for dim, npd in ((2, 32), (3, 10)):
        nn, nc, cnt = orc.fem_sizes(dim, npd)
        I, J, V = orc.fem_stream(dim, npd, seed=0x5EED0004, order_mode=1)
        A = orc.ExtendableSparseMatrix(nn, nn)
        A.apply(np.full(cnt, RAW, np.uint8), I, J, V)
        cp, rv, nz = A.arrays()
        tag = "fem%dd_%d" % (dim, npd)
        fem[tag + "_colptr"], fem[tag + "_rowval"], fem[tag + "_nzval"] = cp, rv, nz
        lines.append("%s nnz=%d csc=%s stream=%s" % (tag, len(rv), digest(cp, rv, nz), digest(I, J, V)))
    np.savez_compressed(os.path.join(HERE, "fem_small.npz"), **fem)
    with open(os.path.join(HERE, "digests.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
