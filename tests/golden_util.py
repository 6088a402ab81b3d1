import hashlib
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STREAMS = ("updates_trace", "assembly_a", "assembly_b", "assembly_c")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def digest(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def cfg3_new_positions(n, seed=0x5EED0003):
    """BASELINE config 3's new entries on an n^3 stencil: the x second-neighbour pairs (l,l+2),(l+2,l),
    values U[0,1) from numpy's default_rng(seed) (host-made, so that the oracle and the device see the same bits)."""
    g = np.arange(n ** 3, dtype=np.int64)
    l = g[(g % n) < n - 2] + 1
    v = np.random.default_rng(seed).random(len(l))
    return np.concatenate([l, l + 2]), np.concatenate([l + 2, l]), np.concatenate([v, v])


def mt_per_entry_streams(n=4000000, p=16, cnt=2000000, seed=1):
    """The per-entry form of the reference's multi-threaded assembly (test/femtools.jl:88-107: every task calls updateindex! /
    rawupdateindex! with its tid): task t sends cnt calls to the columns of its band (columns ascending, rows within 30 of the
    column, kinds UPDATE / RAWUPDATE mixed).  Returns [(I, J, V, K)] per task; bench.py's cfg_mt_sum_per_entry and its pin."""
    rng = np.random.default_rng(seed)
    out = []
    for t in range(p):
        J = np.sort(rng.integers(t * n // p + 1, (t + 1) * n // p + 1, cnt))
        I = np.clip(J + rng.integers(-30, 31, cnt), 1, n)
        out.append((I, J, rng.standard_normal(cnt), rng.integers(1, 3, cnt).astype(np.uint8)))
    return out


def digests(name="digests.txt"):
    out = {}
    with open(os.path.join(GOLDEN, name)) as f:
        for line in f:
            parts = line.split()
            out[parts[0]] = dict(p.split("=") for p in parts[1:])
    return out


def replay(fx, make_matrix, apply, flush, arrays):
    """Replays a stream fixture; yields (flush index, produced arrays, expected arrays, rebuilt)."""
    A = make_matrix(int(fx["m"]), int(fx["n"]))
    prev = 0
    for q, p in enumerate(fx["flush_after"]):
        p = int(p)
        apply(A, fx["kinds"][prev:p], fx["I"][prev:p], fx["J"][prev:p], fx["V"][prev:p])
        flush(A)
        yield q, arrays(A), (fx["colptr%d" % q], fx["rowval%d" % q], fx["nzval%d" % q]), int(fx["rebuilt%d" % q][0])
        prev = p
