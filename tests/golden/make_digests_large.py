"""Full-size parity pins: sha256 digests of the ORACLE's CSC for the bench configurations, committed as
tests/golden/digests_large.txt (the GPU tests digest the device arrays and compare).

Run:  python tests/golden/make_digests_large.py        (about 8 GB of host memory, a minute or two)
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
from golden_util import digest, cfg3_new_positions  # noqa: E402

UPDATE, RAW = 1, 2


def main():
    lines = []
    # BASELINE config 2 at full size (the bench configuration: rand_mode 1, seed 0x5EED0002, updateindex! style) and
    # two smaller cubes
    for n in (128, 192, 256):
        O = orc.fdrand(n, n, n, rand_mode=1, seed=0x5EED0002, style=orc.KIND_UPDATE)
        cp, rv, nz = O.arrays()
        lines.append("fd_%d_m1 nnz=%d csc=%s" % (n, len(rv), digest(cp, rv, nz)))
        print(lines[-1], flush=True)
        if n == 128:
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
    for dim, npd in ((2, 1000), (3, 64)):
        for order in (0, 1):
            nn, nc, cnt = orc.fem_sizes(dim, npd)
            I, J, V = orc.fem_stream(dim, npd, seed=0x5EED0004, order_mode=order)
            A = orc.ExtendableSparseMatrix(nn, nn)
            A.apply(np.full(cnt, RAW, np.uint8), I, J, V)
            cp, rv, nz = A.arrays()
            lines.append("fem%dd_%d_o%d nnz=%d csc=%s" % (dim, npd, order, len(rv), digest(cp, rv, nz)))
            print(lines[-1], flush=True)
            del A, I, J, V
    with open(os.path.join(HERE, "digests_large.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
