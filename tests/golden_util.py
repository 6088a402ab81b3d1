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
