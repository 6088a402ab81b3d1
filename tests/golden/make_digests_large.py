"""Full-size parity pins: sha256 digests of the ORACLE's CSC for the bench configurations, committed as
tests/golden/digests_large.txt (the GPU tests digest the device arrays and compare).

Run:  python tests/golden/make_digests_large.py [tag-prefix ...]   (all, or only the lines whose tag starts with a prefix;
                                                                     lines of other tags are kept from the file)
The bench-size pins (cfg3_256, fem2d_3163, fem3d_216: what bench.py's extra.configs time) take about 20 GB of host
memory and a quarter of an hour; the FEM streams are fed to the oracle in chunks of cells (orc_fem_stream_range).
Like make_golden.py these are oracle outputs (the Julia reference cannot run here).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
from oracle import oracle as orc  # noqa: E402
from golden_util import digest, cfg3_new_positions, mt_per_entry_streams  # noqa: E402

UPDATE, RAW = 1, 2


def fem_chunked(dim, npd, order, chunk_cells=1 << 22):
    """P1 FEM through the oracle's ExtendableSparseMatrix, the stream generated and applied chunk by chunk."""
    nn, nc, cnt = orc.fem_sizes(dim, npd)
    A = orc.ExtendableSparseMatrix(nn, nn)
    for p0 in range(0, nc, chunk_cells):
        p1 = min(nc, p0 + chunk_cells)
        I, J, V = orc.fem_stream_range(dim, npd, p0, p1, seed=0x5EED0004, order_mode=order)
        A.apply(np.full(len(I), RAW, np.uint8), I, J, V)
        del I, J, V
    return A


def elements_chunked(dim, npd, node_mode, chunk_cells=1 << 22):
    """The same assembly as a caller with a mesh in memory runs it (test/femtools.jl:61-69 over cellnodes / elmat / diag:
    orc_fem_mesh_range + orc_elements_stream), random cell order, nodes in natural (0) or permuted (1) numbering."""
    nn, nc, cnt = orc.fem_sizes(dim, npd)
    A = orc.ExtendableSparseMatrix(nn, nn)
    for p0 in range(0, nc, chunk_cells):
        p1 = min(nc, p0 + chunk_cells)
        cn, em, dg = orc.fem_mesh(dim, npd, seed=0x5EED0004, order_mode=1, node_mode=node_mode, node_seed=0x5EED0014, p0=p0, p1=p1)
        I, J, V = orc.elements_stream(cn, em, dg)
        A.apply(np.full(len(I), RAW, np.uint8), I, J, V)
        del I, J, V, cn, em, dg
    return A


def main():
    only = sys.argv[1:]
    want = lambda tag: not only or any(tag.startswith(p) for p in only)  # noqa: E731
    path = os.path.join(HERE, "digests_large.txt")
    kept = {}
    if only and os.path.exists(path):
        with open(path) as f:
            for ln in f:
                if ln.strip():
                    kept[ln.split()[0]] = ln.strip()
    lines = []
    # BASELINE config 2 at full size (the bench configuration: rand_mode 1, seed 0x5EED0002, updateindex! style) and
    # two smaller cubes; 322^3: 33 key bits below the planned prefix (the FINE partition: 4-byte keys above 32 bits, unsharded and as
    # a one-rank shard -- bench.py's cfg_large_322)
    for n in (128, 192, 256, 322):
        if not (want("fd_%d_m1" % n) or want("cfg3_%d" % n)):
            continue
        O = orc.fdrand(n, n, n, rand_mode=1, seed=0x5EED0002, style=orc.KIND_UPDATE)
        cp, rv, nz = O.arrays()
        if want("fd_%d_m1" % n):
            lines.append("fd_%d_m1 nnz=%d csc=%s" % (n, len(rv), digest(cp, rv, nz)))
            print(lines[-1], flush=True)
        if n in (128, 256) and want("cfg3_%d" % n):
            # BASELINE config 3 at 128^3: the stored stencil + new x second-neighbour positions + the full stream again
            I2, J2, V2 = cfg3_new_positions(n)
            O.apply(np.full(len(I2), UPDATE, np.uint8), I2, J2, V2)
            I, J, V = orc.fdrand_stream(n, n, n, rand_mode=1, seed=0x5EED0012)
            O.apply(np.full(len(I), UPDATE, np.uint8), I, J, V)
            O.flush()
            cp, rv, nz = O.arrays()
            lines.append("cfg3_%d nnz=%d csc=%s" % (n, len(rv), digest(cp, rv, nz)))
            print(lines[-1], flush=True)
        del O, cp, rv, nz
    # BASELINE config 4: P1 FEM in random and in natural cell order
    # (3163^2 and 216^3 in random order: the sizes bench.py's extra.configs time)
    for dim, npd, orders in ((2, 1000, (0, 1)), (3, 64, (0, 1)), (2, 3163, (1,)), (3, 216, (1,))):
        for order in orders:
            tag = "fem%dd_%d_o%d" % (dim, npd, order)
            if not want(tag):
                continue
            A = fem_chunked(dim, npd, order)
            cp, rv, nz = A.arrays()
            lines.append("%s nnz=%d csc=%s" % (tag, len(rv), digest(cp, rv, nz)))
            print(lines[-1], flush=True)
            del A, cp, rv, nz
    # ... and from element arrays whose nodes carry a PERMUTED numbering (esp_append_elements: nothing may lean on grid arithmetic)
    for dim, npd in ((2, 1000), (3, 64), (2, 3163), (3, 216)):
        tag = "elem%dd_%d_p1" % (dim, npd)
        if not want(tag):
            continue
        A = elements_chunked(dim, npd, 1)
        cp, rv, nz = A.arrays()
        lines.append("%s nnz=%d csc=%s" % (tag, len(rv), digest(cp, rv, nz)))
        print(lines[-1], flush=True)
        del A, cp, rv, nz
    # flush! of the MT wrapper (genericmtextendablesparsematrixcsc.jl:45-51 = Base.sum(xmatrices, csc)): the 2-D mesh in its natural
    # cell order dealt to 16 partition buffers (tid t = the cells [nc t / 16, nc (t + 1) / 16): a band of the grid), every task's
    # loop of test/femtools.jl:88-107 with rawupdateindex!, then ONE flush! -- bench.py's cfg_mt_sum and the test of the same name
    for npd, p in ((1000, 16), (3163, 16)):
        tag = "mt2d_%d_p%d" % (npd, p)
        if not want(tag):
            continue
        nn, nc, cnt = orc.fem_sizes(2, npd)
        M = orc.MTExtendableSparseMatrix(nn, nn, p)
        for t in range(p):
            a, b = nc * t // p, nc * (t + 1) // p
            for p0 in range(a, b, 1 << 22):
                p1 = min(b, p0 + (1 << 22))
                cn, em, dg = orc.fem_mesh(2, npd, seed=0x5EED0004, order_mode=0, node_mode=0, p0=p0, p1=p1)
                I, J, V = orc.elements_stream(cn, em, dg)
                M.apply(None, I, J, V, tid=t + 1)
                del I, J, V, cn, em, dg
        M.flush()
        cp, rv, nz = M.arrays()
        lines.append("%s nnz=%d csc=%s" % (tag, len(rv), digest(cp, rv, nz)))
        print(lines[-1], flush=True)
        del M, cp, rv, nz
    # ... and its per-entry form: 16 tasks x 2 10^6 updateindex! / rawupdateindex! calls with their tid, ONE flush! (bench.py's
    # cfg_mt_sum_per_entry: esp_flush_sum's general path, the buffers' folds side by side)
    if want("mtgen_4M_p16"):
        n, p = 4000000, 16
        M = orc.MTExtendableSparseMatrix(n, n, p)
        for t, (I, J, V, K) in enumerate(mt_per_entry_streams(n, p)):
            M.apply(K, I, J, V, tid=t + 1)
        M.flush()
        cp, rv, nz = M.arrays()
        lines.append("mtgen_4M_p16 nnz=%d csc=%s" % (len(rv), digest(cp, rv, nz)))
        print(lines[-1], flush=True)
        del M, cp, rv, nz
    for ln in lines:
        kept[ln.split()[0]] = ln
    with open(path, "w") as f:
        f.write("\n".join(kept.values() if only else lines) + "\n")


if __name__ == "__main__":
    main()
